#!/usr/bin/env python3
"""Combine the three rocprofv3 passes of tools/roofline_kernel.py into a traffic / roofline table.
usage: pmc_report.py <out dir holding trace/ fetch/ write/>"""
import csv, glob, json, sys, collections
d = sys.argv[1]


def counter(sub, name):
    f = glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def durations():
    f = glob.glob(f"{d}/trace/**/*kernel_trace.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc


fetch, write, dur = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE"), durations()
med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
# calibration on the copy kernel: 256 MiB read + 256 MiB written per launch; counters are in KiB-like units of 1024 B
cal_name = next(k for k in fetch if "copy" in k.lower() or "Copy" in k)
known = 256 * 1024 * 1024
cf = known / (med(fetch[cal_name]) * 1024)
cw = known / (med(write[cal_name]) * 1024)
out = {"calibration": {"kernel": cal_name[:60], "fetch_factor": round(cf, 3), "write_factor": round(cw, 3)}}
# SURVEY 8d algorithmic bytes per launch: stand-alone scatter-aggregate (20 E' + 8 R F per graph), fused stack lower
# bound (4 R H0 + 20 E + 4 R D per graph), 512 graphs; dense stress shape, 64 graphs
alg = {"k_gcn_propagate_fwd_q": 512 * 16920, "k_sgcn_stack_fwd": 512 * 18000, "k_sgcn_front_fwd": 512 * 18000,
       "k_gcn_propagate_fwd_lds": 64 * (20 * 262144 + 8 * 512 * 16), "k_ds_agg": 64 * (20 * 262144 + 8 * 512 * 16),
       "k_ds_aggT": 64 * (20 * 262144 + 8 * 512 * 16), "k_ds_mask_bwd": 32 * (24 * 262144), "k_ds_deg": 64 * 24 * 262144}
for k in fetch:
    for tag, ab in alg.items():
        if tag + "<" in k or tag + "(" in k:
            fb, wb = med(fetch[k]) * 1024 * cf, med(write[k]) * 1024 * cw
            us = med(dur[k])
            out[tag] = {"us_median": round(us, 2), "alg_bytes": ab, "hbm_read_bytes": int(fb), "hbm_write_bytes": int(wb),
                        "traffic_bytes": int(fb + wb), "achieved_alg_GBps": round(ab / us / 1e3, 1),
                        "frac_of_8TBps": round(ab / us / 1e3 / 8000, 4)}
print(json.dumps(out, indent=1))
