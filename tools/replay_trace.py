#!/usr/bin/env python3
"""Replay-only kernel trace of the graphed train step.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 <repo>/tools/replay_trace.py [workload] [replays]
    python3 tools/replay_trace.py --summarise <dir> [<out.csv>]

Run under the profiler this script builds the step (warm-up + capture), then brackets K graph REPLAYS between two
launches of a marker kernel (k_launch_floor, which no step contains).  ``--summarise`` reads the profiler's
kernel_trace.csv, keeps the dispatches between the markers and prints / writes per-kernel launches and microseconds
PER REPLAY — the warm-up's eager steps, the capture and the model set-up are outside the bracket, so "launches per
replay" is read straight off the file."""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARK = "k_launch_floor"


def summarise(d, out=None):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if MARK in r["Kernel_Name"]]
    assert len(marks) >= 2, "marker launches not found"
    body = rows[marks[-2] + 1:marks[-1]]
    meta = glob.glob(os.path.join(d, "**", "replay_meta.txt"), recursive=True)
    k = int(open(meta[0]).read().split()[0]) if meta else 1
    agg = {}
    for r in body:
        n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(body[-1]["End_Timestamp"]) - int(body[0]["Start_Timestamp"])) / 1e3 / k
    tot_l = sum(a[0] for a in agg.values()) / k
    tot_t = sum(a[1] for a in agg.values()) / k
    lines = [("kernel", "launches_per_replay", "us_per_replay", "avg_us")]
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append((n, f"{c / k:.2f}", f"{t / k:.2f}", f"{t / c:.2f}"))
    lines.append(("TOTAL", f"{tot_l:.2f}", f"{tot_t:.2f}", ""))
    lines.append(("SPAN first start -> last end per replay (us)", "", f"{span:.2f}", ""))
    torch_l = sum(c for n, (c, t) in agg.items() if not n.startswith("k_")) / k
    torch_t = sum(t for n, (c, t) in agg.items() if not n.startswith("k_")) / k
    lines.append(("non-libigcn (torch copies / adds / fills)", f"{torch_l:.2f}", f"{torch_t:.2f}", ""))
    if out:
        with open(out, "w", newline="") as fh:
            csv.writer(fh).writerows(lines)
        # the dispatches of the LAST replay, in order: start offset and duration of every kernel of one step
        per = len(body) // k
        one = body[-per:] if per * k == len(body) else []
        if one:
            t0 = int(one[0]["Start_Timestamp"])
            with open(out.replace(".csv", "_one_replay.csv"), "w", newline="") as fh:
                w = csv.writer(fh)
                # launch geometry and per-workgroup resources as the profiler reports them: what decides which launches
                # can share a grid as two roles (block size, LDS and registers of a paired kernel are the larger of the two)
                w.writerow(("index", "kernel", "start_us", "duration_us", "workgroups", "threads", "lds_bytes", "vgpr",
                            "agpr", "sgpr", "scratch"))
                g = lambda r, k_: r.get(k_, "")                                        # noqa: E731
                for i, r in enumerate(one):
                    wg = int(g(r, "Workgroup_Size_X") or 1) * int(g(r, "Workgroup_Size_Y") or 1) * int(g(r, "Workgroup_Size_Z") or 1)
                    gr = int(g(r, "Grid_Size_X") or 1) * int(g(r, "Grid_Size_Y") or 1) * int(g(r, "Grid_Size_Z") or 1)
                    w.writerow((i, re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0][:100],
                                f"{(int(r['Start_Timestamp']) - t0) / 1e3:.2f}",
                                f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.2f}",
                                gr // max(wg, 1), wg, g(r, "LDS_Block_Size"), g(r, "VGPR_Count"), g(r, "Accum_VGPR_Count"),
                                g(r, "SGPR_Count"), g(r, "Scratch_Size")))
    for ln in lines:
        print(f"{ln[0][:90]:90s} {ln[1]:>8s} {ln[2]:>10s} {ln[3]:>8s}")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        return summarise(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    workload = sys.argv[1] if len(sys.argv) > 1 else "full"
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from igcn_amd import synth
    from igcn_amd._lib import call, stream_ptr
    from igcn_amd.data import Batch
    from igcn_amd.train import FlatAdam, GraphedTrainStep
    dev = torch.device("cuda", 0)
    wl = bench.WORKLOADS[workload]
    # IGCN_TRACE_MODEL="layers,hidden": another entry of the reference's sweep (main.py:152-154) instead of (2, 16)
    lh = [int(v) for v in os.environ.get("IGCN_TRACE_MODEL", "0,0").split(",")]
    model, _ = bench.build_model(dev, wl, layers=lh[0] or None, hidden=lh[1] or None)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    data = Batch.from_data_list(synth.brain_graph_list(wl["graphs"], seed=1000, rois=wl["rois"], tsne_dim=90,
                                                       dense=wl["dense"])).to(dev)
    data.x.requires_grad_(True)
    step = GraphedTrainStep(model, opt, data)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    mark = torch.empty(64, 16, device=dev)
    call("igcn_launch_floor", 64, 16, 0, 0, mark.data_ptr(), stream_ptr())
    for _ in range(k):
        step()
    call("igcn_launch_floor", 64, 16, 0, 0, mark.data_ptr(), stream_ptr())
    torch.cuda.synchronize()
    out = os.environ.get("IGCN_REPLAY_META")
    if out:
        with open(out, "w") as fh:
            fh.write(f"{k} replays {workload}\n")
    print(f"{k} replays of {workload}: loss {float(step.loss):.6f}")


if __name__ == "__main__":
    main()
