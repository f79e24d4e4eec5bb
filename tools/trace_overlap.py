"""Reads a rocprofv3 --kernel-trace CSV and prints the launches around the last LARGE hop kernel in it (start and end relative
to the window's first launch, microseconds): shows which launches of a forward overlap.
   tools/trace_overlap.py <kernel_trace.csv> [kernel-name substring = k_hops] [launches before = 8] [after = 4]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
name = sys.argv[2] if len(sys.argv) > 2 else "k_hops"
before = int(sys.argv[3]) if len(sys.argv) > 3 else 8
after = int(sys.argv[4]) if len(sys.argv) > 4 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
hits = [i for i, r in enumerate(rows) if name in r["Kernel_Name"]]
longest = max(dur(rows[i]) for i in hits)
big = [i for i in hits if dur(rows[i]) > longest // 2]
i = big[-2] if len(big) > 1 else big[-1]
win = rows[max(i - before, 0):i + after]
t0 = int(win[0]["Start_Timestamp"])
for r in win:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} .. {e / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  {r['Kernel_Name'][:90]}")
