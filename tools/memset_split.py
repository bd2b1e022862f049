#!/usr/bin/env python3
"""Separate the one-off fills of a semantic-flow profile from the per-keyframe ones (the kernel table counts
`__amd_rocclr_fillBufferAligned` over the whole process - pool zeroing at volume creation next to the two small fills a keyframe
really issues).  Reads a rocprofv3 --kernel-trace csv of tools/bench_semantic.py and splits the fills by where they sit:
inside a keyframe's launch sequence (between a k_shadow_hist<0> and the fold that ends the keyframe) or outside (set-up, queries).
usage: python tools/memset_split.py <dir-or-csv>   -> one JSON line"""
import csv
import glob
import json
import os
import sys


def main():
    path = sys.argv[1]
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: r["Kernel_Name"].split("(")[0]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    inside, outside, keyframes, launches = [], [], 0, []
    in_kf, n_in_kf = False, 0
    for r in rows:
        nm = name(r)
        if "k_shadow_hist<0>" in nm:  # a keyframe's first launch
            in_kf, n_in_kf = True, 0
            keyframes += 1
        is_fill = "fillBuffer" in nm
        if in_kf:
            n_in_kf += 1
            if is_fill:
                inside.append(dur(r))
        elif is_fill:
            outside.append(dur(r))
        if in_kf and "k_semb_fold_tasks" in nm:  # ... and its last
            in_kf = False
            launches.append(n_in_kf)
    # a keyframe's trailing fill (the next call's counters) is issued right after the fold: count the fill that directly follows
    trailing = 0
    for a, b in zip(rows, rows[1:]):
        if "k_semb_fold_tasks" in name(a) and "fillBuffer" in name(b):
            trailing += 1
    out = {
        "keyframes": keyframes,
        "launches_per_keyframe_incl_fills": round(sum(launches) / max(len(launches), 1), 2),
        "fills_inside_keyframes": len(inside),
        "fills_inside_per_keyframe": round(len(inside) / max(keyframes, 1), 2),
        "fills_inside_mean_us": round(sum(inside) / max(len(inside), 1), 2),
        "fills_directly_after_a_keyframe": trailing,
        "fills_outside_keyframes": len(outside),
        "fills_outside_total_ms": round(sum(outside) / 1e3, 3),
        "fills_outside_max_us": round(max(outside), 1) if outside else 0.0,
        "what": "inside = between a keyframe's first launch (k_shadow_hist<0>) and its last (k_semb_fold_tasks); outside = volume set-up (pool / table / occupancy zeroing), get_voxels / get_object_segments queries and the fill that follows a keyframe",
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
