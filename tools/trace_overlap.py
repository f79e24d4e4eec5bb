"""Reads a rocprofv3 --kernel-trace CSV and prints, for the last forward in it, each kernel's start and end relative to the first
(microseconds): shows which launches of a split batch overlap.   tools/trace_overlap.py <kernel_trace.csv> [n_last_kernels]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} .. {e / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  {r['Kernel_Name'][:90]}")
