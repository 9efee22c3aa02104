#!/usr/bin/env python3
"""HBM traffic of every kernel of the graphed train step, per replay.

Three rocprofv3 passes over tools/replay_trace.py (tools/step_traffic.sh): a kernel trace for the durations, and
``--pmc FETCH_SIZE`` / ``--pmc WRITE_SIZE`` in passes of their own (MI355X_MICROARCH.md: counters never share a run with a
trace) for the bytes.  The replays are bracketed by two marker launches (k_launch_floor) in every pass, so the rows are
REPLAYS only.  Counter units and the gfx950 correction are calibrated the way tools/pmc_report.py does it — a 256 MiB
copy of known size — and read from profiles/<tag>_pmc/traffic.json (``calibration``).

    python3 tools/step_traffic.py <dir with trace/ fetch/ write/> <replays> [calibration.json] > table.csv
"""
import collections
import csv
import glob
import json
import re
import sys

MARK = "k_launch_floor"


def _name(n):
    return re.sub(r"^void ", "", n).split("(")[0][:80]


def _bracket(rows):
    marks = [i for i, r in enumerate(rows) if MARK in r["Kernel_Name"]]
    assert len(marks) >= 2, "marker launches not found"
    return rows[marks[-2] + 1:marks[-1]]


def main():
    d, k = sys.argv[1], int(sys.argv[2])
    cal = {"fetch_factor": 2.0, "write_factor": 1.0}
    if len(sys.argv) > 3:
        cal = json.load(open(sys.argv[3]))["calibration"]
    tr = list(csv.DictReader(open(glob.glob(f"{d}/trace/**/*kernel_trace.csv", recursive=True)[0])))
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    body = _bracket(tr)
    for r in body:
        a = dur[_name(r["Kernel_Name"])]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    byt = {}
    for sub, cname, f in (("fetch", "FETCH_SIZE", cal["fetch_factor"]), ("write", "WRITE_SIZE", cal["write_factor"])):
        rows = [r for r in csv.DictReader(open(glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True)[0]))
                if r["Counter_Name"] == cname]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        acc = collections.defaultdict(float)
        for r in _bracket(rows):
            acc[_name(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1024 * f
        byt[sub] = acc
    w = csv.writer(sys.stdout)
    w.writerow(("kernel", "launches_per_replay", "us_per_replay", "MB_read_per_replay", "MB_written_per_replay",
                "TB_per_s", "frac_of_8TBps"))
    tot = [0.0, 0.0, 0.0, 0.0]
    for n, (c, t) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        rd, wr = byt["fetch"].get(n, 0.0) / k, byt["write"].get(n, 0.0) / k
        us = t / k
        rate = (rd + wr) / us / 1e6 if us else 0.0
        w.writerow((n, f"{c / k:.2f}", f"{us:.2f}", f"{rd / 1e6:.2f}", f"{wr / 1e6:.2f}", f"{rate:.2f}", f"{rate / 8:.3f}"))
        tot[0] += c / k; tot[1] += us; tot[2] += rd; tot[3] += wr
    rate = (tot[2] + tot[3]) / tot[1] / 1e6
    w.writerow(("TOTAL", f"{tot[0]:.2f}", f"{tot[1]:.2f}", f"{tot[2] / 1e6:.2f}", f"{tot[3] / 1e6:.2f}", f"{rate:.2f}",
                f"{rate / 8:.3f}"))


if __name__ == "__main__":
    main()
