#!/usr/bin/env python3
"""Per-kernel durations and the idle gap before each kernel, from a rocprofv3 --kernel-trace csv."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-28:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append(e - s)
    if prev_end is not None:
        gap[name].append(s - prev_end)
    prev_end = max(e, prev_end or e)
for k in dur:
    d = sorted(dur[k]); g = sorted(gap[k]) or [0]
    print(f"{k:30s} n={len(d):4d} dur med {d[len(d)//2]/1e3:8.2f} us   gap-before med {g[len(g)//2]/1e3:7.2f} us")
