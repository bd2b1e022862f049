// libpyslam_hipvol.so — per-frame image preparation on the GPU: cv2.remap as pySLAM uses it for
// undistortion (pyslam/dense/volumetric_integrator_base.py:1017-1043): colour INTER_LINEAR, depth and
// label images INTER_NEAREST, float32 maps, BORDER_CONSTANT 0.
//
// OpenCV semantics restated (OpenCV is not available in this image: parity with cv2 is UNPINNED):
//   nearest : src(cvRound(map_y), cvRound(map_x)), cvRound = round-half-to-even
//   linear  : fixed-point bilinear, INTER_BITS = 5: sx = cvRound(map_x * 32), ix = sx >> 5, fx = sx & 31;
//             8-bit images use integer weights (32-fx)(32-fy)*32 ... summing to 2^15 and
//             (sum + 2^14) >> 15; float images use the same 1/32-quantised weights in float.
#include "hv_common.h"

enum { HV_IMG_U8 = 0, HV_IMG_F32 = 1, HV_IMG_I32 = 2 };

template <typename T> __device__ __forceinline__ T px_or_zero(const T *src, int H, int W, int C, int y, int x, int c) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? src[((int64_t)y * W + x) * C + c] : (T)0;
}

template <typename T>
__global__ __launch_bounds__(256) void k_remap_nearest(const T *__restrict__ src, int H, int W, int C,
                                                        const float *__restrict__ mx, const float *__restrict__ my,
                                                        T *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int sx = __float2int_rn(mx[i]), sy = __float2int_rn(my[i]);
    for (int c = 0; c < C; ++c) dst[i * C + c] = px_or_zero(src, H, W, C, sy, sx, c);
}

__global__ __launch_bounds__(256) void k_remap_linear_u8(const uint8_t *__restrict__ src, int H, int W, int C,
                                                          const float *__restrict__ mx, const float *__restrict__ my,
                                                          uint8_t *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int sx = __float2int_rn(mx[i] * 32.0f), sy = __float2int_rn(my[i] * 32.0f);
    const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    for (int c = 0; c < C; ++c) {
        const int s = (int)px_or_zero(src, H, W, C, iy, ix, c) * w00 + (int)px_or_zero(src, H, W, C, iy, ix + 1, c) * w01 +
                      (int)px_or_zero(src, H, W, C, iy + 1, ix, c) * w10 + (int)px_or_zero(src, H, W, C, iy + 1, ix + 1, c) * w11;
        const int r = (s + (1 << 14)) >> 15;
        dst[i * C + c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}

__global__ __launch_bounds__(256) void k_remap_linear_f32(const float *__restrict__ src, int H, int W, int C,
                                                           const float *__restrict__ mx, const float *__restrict__ my,
                                                           float *__restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int sx = __float2int_rn(mx[i] * 32.0f), sy = __float2int_rn(my[i] * 32.0f);
    const int ix = sx >> 5, iy = sy >> 5;
    const float fx = (float)(sx & 31) * (1.0f / 32.0f), fy = (float)(sy & 31) * (1.0f / 32.0f);
    for (int c = 0; c < C; ++c) {
        dst[i * C + c] = px_or_zero(src, H, W, C, iy, ix, c) * ((1.0f - fx) * (1.0f - fy)) +
                         px_or_zero(src, H, W, C, iy, ix + 1, c) * (fx * (1.0f - fy)) +
                         px_or_zero(src, H, W, C, iy + 1, ix, c) * ((1.0f - fx) * fy) +
                         px_or_zero(src, H, W, C, iy + 1, ix + 1, c) * (fx * fy);
    }
}

// The reference's per-keyframe pair of remaps (colour INTER_LINEAR, depth INTER_NEAREST: volumetric_integrator_base.py:1017-1043) for
// a whole batch of device-resident frames in ONE launch: a thread owns an output pixel, reads its map entry once and samples every
// frame of the batch with it.  Depth keeps its storage type (float32, or the sensor's uint16: the nearest-neighbour pick commutes with
// the later conversion to metres), so a TUM-style keyframe stays 5 bytes per pixel on its way through the GPU.
template <typename D>
__global__ __launch_bounds__(256) void k_rectify_frames(const D *__restrict__ depth, const uint8_t *__restrict__ rgb, int n_frames, int H, int W,
                                                         const float *__restrict__ mx, const float *__restrict__ my,
                                                         D *__restrict__ depth_out, uint8_t *__restrict__ rgb_out) {
    const int64_t npx = (int64_t)H * W;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    const float fxm = mx[i], fym = my[i];
    const int nx = __float2int_rn(fxm), ny = __float2int_rn(fym);
    const int sx = __float2int_rn(fxm * 32.0f), sy = __float2int_rn(fym * 32.0f);
    const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    for (int f = 0; f < n_frames; ++f) {
        const D *df = depth + (int64_t)f * npx;
        const uint8_t *cf = rgb + (int64_t)f * npx * 3;
        depth_out[(int64_t)f * npx + i] = px_or_zero(df, H, W, 1, ny, nx, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int s = (int)px_or_zero(cf, H, W, 3, iy, ix, c) * w00 + (int)px_or_zero(cf, H, W, 3, iy, ix + 1, c) * w01 +
                          (int)px_or_zero(cf, H, W, 3, iy + 1, ix, c) * w10 + (int)px_or_zero(cf, H, W, 3, iy + 1, ix + 1, c) * w11;
            const int r = (s + (1 << 14)) >> 15;
            rgb_out[((int64_t)f * npx + i) * 3 + c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
}

// (hv_tsdf.hip: queued on `s`, nothing waits)
int hv_rectify_frames_device(hv_volume *v, hipStream_t s, const void *d_depth, int32_t depth_dtype, const uint8_t *d_rgb, int n_frames,
                             int height, int width, void *d_depth_out, uint8_t *d_rgb_out) {
    const int64_t npx = (int64_t)height * width;
    const dim3 grid((unsigned)((npx + 255) / 256)), block(256);
    if (depth_dtype == HV_DEPTH_U16)
        hipLaunchKernelGGL(k_rectify_frames<uint16_t>, grid, block, 0, s, (const uint16_t *)d_depth, d_rgb, n_frames, height, width,
                           (const float *)v->rect_map_x, (const float *)v->rect_map_y, (uint16_t *)d_depth_out, d_rgb_out);
    else
        hipLaunchKernelGGL(k_rectify_frames<float>, grid, block, 0, s, (const float *)d_depth, d_rgb, n_frames, height, width,
                           (const float *)v->rect_map_x, (const float *)v->rect_map_y, (float *)d_depth_out, d_rgb_out);
    HV_HIP(hipGetLastError());
    return HV_OK;
}

extern "C" int hv_tsdf_set_rectify_maps(hv_volume *v, const float *map_x, const float *map_y, int32_t height, int32_t width, int32_t loc) {
    HV_REQUIRE(v != nullptr, HV_ERR_INVALID, "hv_tsdf_set_rectify_maps: null volume");
    HV_REQUIRE(v->cfg.mode == HV_MODE_TSDF, HV_ERR_MODE, "hv_tsdf_set_rectify_maps: volume is not in TSDF mode");
    HV_HIP(hipSetDevice(v->device));
    if (v->stream_aux) HV_HIP(hipStreamSynchronize(v->stream_aux));
    HV_HIP(hipStreamSynchronize(v->stream));
    if (map_x == nullptr || map_y == nullptr) { // frames are fused as they come again
        v->rect_W = v->rect_H = 0;
        return HV_OK;
    }
    HV_REQUIRE(height > 0 && width > 0, HV_ERR_INVALID, "hv_tsdf_set_rectify_maps: bad image size");
    const size_t bytes = sizeof(float) * (size_t)height * width;
    void *mxb = v->rect_map_x, *myb = v->rect_map_y;
    int rc = hv_ensure_buffer(v, &mxb, &v->rect_map_x_bytes, bytes);
    if (rc != HV_OK) return rc;
    v->rect_map_x = (float *)mxb;
    rc = hv_ensure_buffer(v, &myb, &v->rect_map_y_bytes, bytes);
    if (rc != HV_OK) return rc;
    v->rect_map_y = (float *)myb;
    HV_HIP(hipMemcpy(v->rect_map_x, map_x, bytes, loc == HV_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    HV_HIP(hipMemcpy(v->rect_map_y, map_y, bytes, loc == HV_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    v->rect_W = width;
    v->rect_H = height;
    return HV_OK;
}

extern "C" int hv_remap(hv_volume *v, const void *src, int32_t src_kind, int32_t channels, int32_t height, int32_t width,
                        const float *map_x, const float *map_y, int32_t linear, void *dst, int32_t loc) {
    HV_REQUIRE(v != nullptr && src != nullptr && map_x != nullptr && map_y != nullptr && dst != nullptr, HV_ERR_INVALID,
               "hv_remap: null argument");
    HV_REQUIRE(height > 0 && width > 0 && channels >= 1 && channels <= 4, HV_ERR_INVALID, "hv_remap: bad image shape");
    HV_REQUIRE(src_kind == HV_IMG_U8 || src_kind == HV_IMG_F32 || src_kind == HV_IMG_I32, HV_ERR_INVALID,
               "hv_remap: unsupported image dtype");
    HV_REQUIRE(!(linear && src_kind == HV_IMG_I32), HV_ERR_INVALID, "hv_remap: label images are remapped with INTER_NEAREST");
    HV_HIP(hipSetDevice(v->device));
    const int64_t npx = (int64_t)height * width;
    const size_t esz = src_kind == HV_IMG_U8 ? 1 : 4;
    const size_t img_bytes = esz * channels * npx, map_bytes = sizeof(float) * npx;
    const void *d_src = src;
    const float *d_mx = map_x, *d_my = map_y;
    void *d_dst = dst;
    if (loc == HV_HOST) {
        int rc = hv_ensure_buffer(v, &v->stage_b, &v->stage_b_bytes, 2 * img_bytes + 2 * map_bytes + 1024);
        if (rc != HV_OK) return rc;
        char *base = (char *)v->stage_b;
        float *mx = (float *)base, *my = mx + npx;
        char *s = (char *)(my + npx);
        s += (256 - ((uintptr_t)s & 255)) & 255;
        char *d = s + ((img_bytes + 255) & ~(size_t)255);
        bool pinned_src = false; // (page-locked sources are read by the DMA engine after hipMemcpyAsync returns: hv_h2d)
        if ((rc = hv_h2d_lazy(v, mx, map_x, map_bytes, &pinned_src)) != HV_OK) return rc;
        if ((rc = hv_h2d_lazy(v, my, map_y, map_bytes, &pinned_src)) != HV_OK) return rc;
        if ((rc = hv_h2d_lazy(v, s, src, img_bytes, &pinned_src)) != HV_OK) return rc;
        if ((rc = hv_h2d_fence(v, pinned_src)) != HV_OK) return rc;
        d_src = s; d_mx = mx; d_my = my; d_dst = d;
    }
    const dim3 grid((unsigned)((npx + 255) / 256)), block(256);
    if (!linear) {
        if (src_kind == HV_IMG_U8)
            hipLaunchKernelGGL(k_remap_nearest<uint8_t>, grid, block, 0, v->stream, (const uint8_t *)d_src, height, width, channels, d_mx, d_my, (uint8_t *)d_dst);
        else if (src_kind == HV_IMG_F32)
            hipLaunchKernelGGL(k_remap_nearest<float>, grid, block, 0, v->stream, (const float *)d_src, height, width, channels, d_mx, d_my, (float *)d_dst);
        else
            hipLaunchKernelGGL(k_remap_nearest<int32_t>, grid, block, 0, v->stream, (const int32_t *)d_src, height, width, channels, d_mx, d_my, (int32_t *)d_dst);
    } else if (src_kind == HV_IMG_U8) {
        hipLaunchKernelGGL(k_remap_linear_u8, grid, block, 0, v->stream, (const uint8_t *)d_src, height, width, channels, d_mx, d_my, (uint8_t *)d_dst);
    } else {
        hipLaunchKernelGGL(k_remap_linear_f32, grid, block, 0, v->stream, (const float *)d_src, height, width, channels, d_mx, d_my, (float *)d_dst);
    }
    HV_HIP(hipGetLastError());
    if (loc == HV_HOST) HV_HIP(hipMemcpyAsync(dst, d_dst, img_bytes, hipMemcpyDeviceToHost, v->stream));
    HV_HIP(hipStreamSynchronize(v->stream));
    return HV_OK;
}
