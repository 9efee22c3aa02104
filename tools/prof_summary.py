#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV directory: per-step totals by owner + top kernels.
usage: prof_summary.py <dir with *_kernel_stats.csv> <steps profiled> [top]"""
import csv, glob, re, sys
d, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = list(csv.DictReader(open(glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0])))
grp = {"igcn": 0.0, "rocblas": 0.0, "torch/other": 0.0}
cnt = dict.fromkeys(grp, 0.0)
for r in rows:
    n, t, c = r["Name"], float(r["TotalDurationNs"]) / 1e3 / steps, int(r["Calls"]) / steps
    k = "igcn" if re.match(r"(void )?k_", n) else ("rocblas" if n.startswith("Cijk") else "torch/other")
    grp[k] += t; cnt[k] += c
print("per step: " + "  ".join(f"{k}: {grp[k]:.0f} us / {cnt[k]:.0f} launches" for k in grp),
      f" total {sum(grp.values()):.0f} us / {sum(cnt.values()):.0f} launches")
for r in rows[:top]:
    n = re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", r["Name"])[:78]
    print(f"{n:78s} {int(r['Calls'])/steps:6.1f}/step {float(r['TotalDurationNs'])/1e3/steps:8.1f} us/step avg {float(r['AverageNs'])/1e3:8.2f}")
