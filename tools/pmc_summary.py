#!/usr/bin/env python3
"""Mean per-dispatch PMC values per kernel from rocprofv3 csv output dirs (pmc_counter_collection.csv).

  tools/pmc_summary.py DIR [DIR ...]                      one text line per (dir, kernel)
  tools/pmc_summary.py --json OUT --command-key KEY [--range SUBSTR:LO:HI] DIR [DIR ...]
        merge the passes into one machine-readable file (bench.py parses it at run time): per kernel the mean
        per-launch value of every counter, the number of launches, the library build digest
        (pyslam_amd/lib/.build_digest) and git revision the passes ran on, and the bench command they profiled.
        --range: for kernels whose name contains SUBSTR keep only launches LO .. HI-1 (dispatch order) - the timed
        steps of bench.py, without its warm-up and secondary legs.
        --also SUBSTR:LO:HI:SUFFIX: a second entry "<kernel name><SUFFIX>" with the means over launches LO .. HI-1 of the
        kernels whose name contains SUBSTR (bench.py's extraction leg: the full passes of a kernel beside its incremental ticks).
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(d, ranges, also=()):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r.get("Dispatch_Id", 0)))
        for r in rows:
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in list(agg):
        for sub, lo, hi, suffix in also:
            if sub in k:
                for c, vals in agg[k].items():
                    picked = [vals[i] for i in range(lo, hi) if i < len(vals)]
                    if picked:
                        agg[k + suffix][c] = picked
    for k, cs in agg.items():
        if any(k.endswith(suffix) for _, _, _, suffix in also):
            continue
        for sub, lo, hi in ranges:
            if sub in k:
                for c in cs:
                    cs[c] = cs[c][lo:hi]
    return agg


def main():
    argv = sys.argv[1:]
    out_json = key = None
    ranges = []
    also = []
    while argv and argv[0].startswith("--"):
        if argv[0] == "--json":
            out_json = argv[1]
        elif argv[0] == "--command-key":
            key = argv[1]
        elif argv[0] == "--range":
            sub, lo, hi = argv[1].rsplit(":", 2)
            ranges.append((sub, int(lo), int(hi)))
        elif argv[0] == "--also":
            sub, lo, hi, suffix = argv[1].rsplit(":", 3)
            also.append((sub, int(lo), int(hi), suffix))
        argv = argv[2:]
    merged = {}
    for d in argv:
        for k, cs in collect(d, ranges, also).items():
            if "fillBuffer" in k or "copyBuffer" in k:
                continue
            n = len(next(iter(cs.values())))
            if n == 0:
                continue
            print(d, k[-40:], " ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(cs.items())), f"n={n}")
            m = merged.setdefault(k, {"launches": n})
            for c, v in cs.items():
                m[c] = sum(v) / len(v)
    if out_json:
        try:
            digest = open(os.path.join(ROOT, "pyslam_amd", "lib", ".build_digest")).read().strip()
        except OSError:
            digest = None
        try:
            git = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        except Exception:
            git = os.environ.get("GRAFT_GIT_HEAD")  # the GPU box has no .git: tools/profile_round.sh passes it in
        json.dump({"build_digest": digest, "git": git, "command_key": key, "units": {"FETCH_SIZE": "KB", "WRITE_SIZE": "KB"},
                   "kernels": merged}, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
