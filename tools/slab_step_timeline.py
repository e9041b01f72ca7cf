"""Prints the last kernels of a rocprofv3 --kernel-trace CSV in time order: start offset, duration, gap to the previous end,
queue, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -70:]
t0 = int(tail[0]["Start_Timestamp"])
prev_end = t0
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:6.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:60]}")
    prev_end = max(prev_end, e)
