#!/usr/bin/env python3
"""Timeline statistics of one steady-state step from a rocprofv3 kernel trace CSV.
usage: trace_gaps.py <dir> [marker kernel substring, default k_plan_segmented]"""
import csv, glob, sys, re, collections
d = sys.argv[1]; marker = sys.argv[2] if len(sys.argv) > 2 else "k_plan_segmented"
rows = list(csv.DictReader(open(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
gaps = [int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"]) for i in range(len(seg) - 1)]
print(f"step span {(t1-t0)/1e3:.1f} us, kernels {len(seg)}, busy {busy/1e3:.1f} us, gaps sum {sum(g for g in gaps if g>0)/1e3:.1f} us, "
      f"median gap {sorted(gaps)[len(gaps)//2]/1e3:.2f} us, overlapped {sum(1 for g in gaps if g<0)}")
big = sorted(((g, i) for i, g in enumerate(gaps)), reverse=True)[:8]
for g, i in big:
    print(f"  gap {g/1e3:7.1f} us after {seg[i]['Kernel_Name'][:60]} -> {seg[i+1]['Kernel_Name'][:50]}")
