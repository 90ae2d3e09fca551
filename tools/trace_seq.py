"""Per-dispatch durations of selected kernels from a rocprofv3 --kernel-trace CSV, in launch order.
usage: trace_seq.py <kernel_trace.csv> <substring> [<substring> ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pats = sys.argv[2:]
prev_end = None
for r in rows:
    name = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    if any(p in name for p in pats):
        print(f"{(e - s) / 1e3:9.1f} us  gap {gap:7.1f} us  grid {r.get('Grid_Size_X', '?'):>8s}  {name[:70]}")
