"""Image decoding with the conventions of the cv2.imread calls in pyslam/io/dataset.py (cv2 is not a dependency
of this package; Pillow does the decoding): colour images come back BGR uint8 [H,W,3]; ``unchanged`` keeps the
file's bit depth (16-bit depth PNGs -> uint16 [H,W])."""
import numpy as np


def _open(path):
    from PIL import Image

    return Image.open(path)


def imread_color(path):
    """cv2.imread(path): BGR uint8 [H,W,3], or None when the file cannot be read."""
    try:
        with _open(path) as im:
            rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    except (OSError, ValueError):
        return None
    return np.ascontiguousarray(rgb[..., ::-1])


def imread_unchanged(path):
    """cv2.imread(path, cv2.IMREAD_UNCHANGED) for single-channel images: uint8 / uint16 / float32 [H,W]
    (colour images: BGR[A] uint8)."""
    try:
        with _open(path) as im:
            if im.mode in ("I;16", "I;16B", "I;16L"):
                return np.ascontiguousarray(np.asarray(im, dtype=np.uint16))
            if im.mode == "I":  # Pillow opens some 16-bit PNGs as 32-bit int
                a = np.asarray(im)
                return np.ascontiguousarray(a.astype(np.uint16) if a.max(initial=0) < 65536 else a.astype(np.int32))
            if im.mode == "F":
                return np.ascontiguousarray(np.asarray(im, dtype=np.float32))
            if im.mode in ("L", "P", "1"):
                return np.ascontiguousarray(np.asarray(im.convert("L"), dtype=np.uint8))
            a = np.asarray(im.convert("RGBA" if "A" in im.mode else "RGB"), dtype=np.uint8)
            return np.ascontiguousarray(a[..., [2, 1, 0, 3]] if a.shape[2] == 4 else a[..., ::-1])
    except (OSError, ValueError):
        return None


def imwrite(path, img):
    """cv2.imwrite for the cases the tests and the dataset writers need (BGR uint8 colour, uint16 / uint8 gray)."""
    from PIL import Image

    a = np.asarray(img)
    if a.ndim == 3:
        Image.fromarray(np.ascontiguousarray(a[..., ::-1])).save(path)
    elif a.dtype == np.uint16:
        Image.fromarray(np.ascontiguousarray(a)).save(path)  # uint16 -> mode I;16
    else:
        Image.fromarray(a.astype(np.uint8)).save(path)
