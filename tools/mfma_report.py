#!/usr/bin/env python3
"""profiles/rNN_pmc/mfma.json from the two rocprofv3 passes of tools/mfma_run.sh (usage: mfma_report.py <dir>).

Per kernel (median over its launches): duration, SQ_VALU_MFMA_BUSY_CYCLES (cycles in which a SIMD's matrix pipe is
busy, summed over the chip's 1024 SIMDs: 32 per v_mfma_f32_16x16x4_f32, 16 per v_mfma_f32_16x16x32_bf16 —
MI355X_MICROARCH.md, per-instruction constants), GRBM_GUI_ACTIVE (sum over the 8 XCDs of the cycles the dispatch kept
the XCD busy) and from them
    mfma_busy_frac = MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
— the fraction of the launch during which the average SIMD's matrix pipe was occupied.  SQ_BUSY_CYCLES (cycles any
wave was resident, summed over the shader engines) and SQ_WAVES are kept beside it.  The 4096^3 product is the
calibration point: its fraction by TIME against the 157.3 TFLOP/s peak is printed next to the counter's."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
SIMDS = 256 * 4
PEAK_F32 = 157.3e12


def counters():
    f = glob.glob(f"{d}/mpmc/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    return acc


def durations():
    f = glob.glob(f"{d}/mtrace/**/*kernel_trace.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc


med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
pmc, dur = counters(), durations()
want = ("k_ds_agg<", "k_ds_aggT<", "k_ds_mask_bwd<", "k_attn_mfma_fwd<", "k_attn_mfma_bwd_shared<", "k_attn_mfma_bwd<",
        "k_attn_split_fwd", "k_attn_split_bwd<",
        "k_attn_bf16_fwd", "k_attn_bf16_bwd_dq", "k_attn_bf16_bwd_dkv", "k_gemm_f32<", "k_proj_fwd", "k_proj_bwd",
        "k_head_bwd", "k_go_attn_bwd_lds<")
out = {"unit": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); medians over launches",
       "kernels": {}}
for k in sorted(pmc):
    short = k[5:] if k.startswith("void ") else k
    short = short.split("(")[0]
    if not any(w in short for w in want):
        continue
    rows = list(pmc[k].values())
    g = lambda name: med([r.get(name, 0.0) for r in rows])      # noqa: E731
    busy, gui = g("SQ_VALU_MFMA_BUSY_CYCLES"), g("GRBM_GUI_ACTIVE")
    us = med(dur[k]) if k in dur else None
    e = {"launches": len(rows), "us_median_unprofiled_pass": round(us, 2) if us else None,
         "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui, "SQ_BUSY_CYCLES": g("SQ_BUSY_CYCLES"),
         "SQ_WAVES": g("SQ_WAVES"),
         "mfma_busy_frac": round(busy / (SIMDS * gui / 8), 4) if gui else None,
         "clock_GHz_during_launch": round(gui / 8 / (us * 1e3), 3) if (gui and us) else None}
    out["kernels"][short + f" #{len(out['kernels'])}"] = e
for k, e in out["kernels"].items():
    if k.startswith("k_gemm_f32<") and e["us_median_unprofiled_pass"] and e["us_median_unprofiled_pass"] > 500:
        e["calibration"] = {"flops": 2 * 4096 ** 3, "frac_of_fp32_peak_by_time":
                            round(2 * 4096 ** 3 / (e["us_median_unprofiled_pass"] * 1e-6) / PEAK_F32, 4)}
print(json.dumps(out, indent=1))
