#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace run (csv): per kernel name the count / mean duration, and for the last N
launches the start / end offsets in microseconds - shows what overlaps what (batch pipeline: touch + pack of batch k+1 beside
the sweep of batch k).  usage: python tools/timeline.py <dir-or-csv> [N]"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    stats = {}
    for r in rows:
        name = r["Kernel_Name"].split("(")[0][:60]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = stats.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += d
    for name, (c, t) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:62s} n={c:5d} mean={t / c:9.1f} us total={t / 1e3:9.2f} ms")
    tail = rows[-n_last:]
    t0 = int(tail[0]["Start_Timestamp"])
    print("--- last launches: start, end (us from the first of them), duration, stream/queue, name")
    for r in tail:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{s:10.1f} {e:10.1f} {e - s:8.1f}  q={r.get('Queue_Id', '?'):>3s}  {r['Kernel_Name'].split('(')[0][:70]}")


if __name__ == "__main__":
    main()
