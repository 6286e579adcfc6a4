#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per kernel and counter."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].replace("rtd::(anonymous namespace)::", "").replace("void ", "")
            name = name.split("(")[0]
            if not name.startswith("k_"):
                continue
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(agg):
            print(k, "calls", len(next(iter(agg[k].values()))))
            for c in sorted(agg[k]):
                v = agg[k][c]
                print(f"    {c:28s} mean {sum(v) / len(v):16.1f}")
