#!/usr/bin/env python3
"""Mean per-dispatch PMC values per kernel from rocprofv3 csv output dirs (pmc_counter_collection.csv)."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0][-40:]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "fillBuffer" in k or "copyBuffer" in k:
                continue
            print(d, k, " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())), f"n={len(next(iter(cs.values())))}")
