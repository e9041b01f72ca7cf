#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean of each counter."""
import collections
import csv
import sys


def summarise(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in paths:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][-64:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


if __name__ == "__main__":
    res = summarise(sys.argv[1:])
    names = sorted({c for d in res.values() for c in d})
    for k, d in res.items():
        if k.startswith("__amd"):
            continue
        print(k)
        for c in names:
            if c in d:
                print("    %-22s %.6g" % (c, d[c]))
