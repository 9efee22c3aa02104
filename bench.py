#!/usr/bin/env python3
"""bench.py — graphs/s of the full IG-GCN train step (SGCN over 90-ROI brain graphs + GO-SNP network).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks 5] [--workload full|sgcn|stress]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    IGCN_BENCH_SWEEP="1,2,4,8" python bench.py ...        one line per N, weak-scaling efficiency vs its own N = 1 run

The timed region (--steps replays between barriers + synchronize) is repeated --blocks times; the line reports the
MEDIAN block and the spread (`timing`).  The default line also carries the loader-fed step (`pipeline`: host feeder
thread / device-resident dataset / device GDC one batch ahead on a side stream) and, at N > 1, the distributed step
taken apart per rank (`distributed_step`).

Default workload (BASELINE.json configs[2]): full SGCN_GCN_IMGSNP (L=2, hidden=16, R=90, H0=3,
cross-attention fusion, 3 classes, 3 regression targets), synthetic GO DAG N=3000 pool [1800,800,300,99,1],
256 graphs per GPU (weak scaling), fp32.  One step = kernel/train_eval_sgcn_img_snps.py:515-547: graph-plan
build, forward, masked forward, 7 loss terms, backward, (all-reduce), Adam — on inputs already resident
in HBM.  Prints ONE JSON line on rank 0.

``--workload stress`` = BASELINE configs[4]: 512-ROI dense graphs + 10k-node GO DAG, 32 graphs per GPU, dense
feature transforms with bf16 operands on the matrix cores (fp32 accumulation); same JSON contract.

Roofline fields (rank 0, N=1):
  roofline        the GCN scatter-aggregate kernel AS THE STEP LAUNCHES IT.  ``us_per_launch`` is the kernel's average
                  duration inside the train step, read from a ``rocprofv3 --kernel-trace --stats`` run of this very
                  command made by this invocation (a child process; the CSV is the one committed under profiles/).
                  ``achieved`` = SURVEY §8d algorithmic bytes / that duration; ``frac`` = achieved / 8 TB/s.
                  Beside it: the bytes the kernel itself moves (8-byte edge records instead of int64 pairs),
                  a hot back-to-back replay and a COLD replay rotating through > 256 MiB of distinct buffers (both
                  timed with HIP events on the launch stream), and the launch floor of the same grid.
  roofline_mfma   dense feature transforms (lin1, K|V projection) against the MFMA peak of their operand type.
  roofline_step   the WHOLE step against the HBM roof: PMC bytes of every kernel of a replay (committed table of
                  tools/step_traffic.sh) / this run's kernel time per replay.
  cpu_baseline    the oracle (CPU restatement of the reference) on this box's host cores, bounded sample.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import igcn_amd  # noqa: E402,F401
from igcn_amd import synth  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402

GRAPHS_PER_GPU = 256
POOL = (1800, 800, 300, 99, 1)
LAYERS, HIDDEN, ROIS = 2, 16, 90
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}     # dense peaks, MI355X_MICROARCH.md § Matrix cores
INFINITY_CACHE_BYTES = 256 << 20

# --workload: the default is the configuration the metric is quoted on
WORKLOADS = {
    "full": dict(rois=ROIS, pool=POOL, graphs=GRAPHS_PER_GPU, dense=False, bf16=False,
                 name="configs[2]: full sgcn_img_snp train step (2 fwd + 7 losses + bwd + Adam), "
                      "90-ROI k=3 brain graphs + 3000-node GO-SNP DAG"),
    "sgcn": dict(rois=ROIS, pool=None, graphs=GRAPHS_PER_GPU, dense=False, bf16=False,
                 name="configs[1]: SGCN-only train step (kernel/train_eval_sgcn.py:296-314), 90-ROI k=3 brain graphs"),
    "stress": dict(rois=512, pool=(6000, 2700, 1000, 299, 1), graphs=32, dense=True, bf16=True,
                   name="configs[4]: full train step, 512-ROI dense brain graphs + 10k-node GO DAG, bf16 feature "
                        "transforms on MFMA (fp32 accumulate)"),
}


def build_model(device, wl=None, bf16=None, layers=None, hidden=None):
    wl = wl or WORKLOADS["full"]
    bf16 = wl["bf16"] if bf16 is None else bf16
    LAYERS, HIDDEN = (layers or globals()["LAYERS"]), (hidden or globals()["HIDDEN"])
    torch.manual_seed(1000)                                   # main.py:102 seed
    if wl["pool"] is None:
        from igcn_amd.sgcn import SGCN_GCN
        model = SGCN_GCN(None, LAYERS, HIDDEN, rois=wl["rois"], H_0=3, num_features=3, num_classes=3).to(device)
        model.train()
        return model, None
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    go_snps, adj, pool_dim = synth.go_hierarchy(wl["pool"], seed=0)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, device)
    model = SGCN_GCN_IMGSNP(LAYERS, HIDDEN, a_g, a, pool_dim, 32, device, rois=wl["rois"], H_0=3, num_classes=3,
                            isSoftSimilarity=True, rbf_gamma=0.01, isCrossAtten=True, num_regr=3,
                            isuseProb4Regr=True, isImageOnly=False, isSNPsOnly=False,
                            bf16_transforms=bf16).to(device)
    model.train()
    return model, (go_snps, adj, pool_dim)


MODEL_SWEEP = [(2, 16), (3, 16), (2, 10), (3, 10), (4, 5)]       # main.py:152-154 (layers, hiddens): the default sweep


def model_sweep(wl, device, data, steps=20, warmup=3):
    """The five (layers, hidden) entries of the reference's own hyper-parameter sweep (main.py:152-154) at the headline
    workload, each as its own captured step: ms per step, graphs/s, and the cost per graph relative to the (2, 16) entry
    the metric is quoted on — how deep the cliff is off the shape the kernels were tuned at."""
    from igcn_amd.train import FlatAdam, GraphedTrainStep
    rows, base = [], None
    for layers, hidden in MODEL_SWEEP:
        model, _ = build_model(device, wl, layers=layers, hidden=hidden)
        opt = FlatAdam(model.parameters(), lr=1e-3)
        d = Batch.from_data_list(synth.brain_graph_list(wl["graphs"], seed=1000, rois=wl["rois"], tsne_dim=90,
                                                        dense=wl["dense"])).to(device)
        d.x.requires_grad_(True)
        try:
            step = GraphedTrainStep(model, opt, d)
        except Exception as exc:                  # noqa: BLE001
            rows.append({"layers": layers, "hidden": hidden, "error": f"{type(exc).__name__}: {exc}"[:200]})
            continue
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / steps)
        ms = sorted(ts)[1] * 1e3
        if base is None:
            base = ms
        rows.append({"layers": layers, "hidden": hidden, "ms_per_step": round(ms, 4),
                     "graphs_per_s": round(wl["graphs"] / ms * 1e3, 1), "cost_per_graph_vs_2x16": round(ms / base, 3),
                     "loss": round(float(step.loss), 5)})
        del step, opt, model
        torch.cuda.empty_cache()
    return {"entries": rows, "reference": "main.py:152-154 (layers, hiddens); same batch, GO DAG and step as the headline",
            "timing": f"median of 3 blocks of {steps} hipGraph replays per entry"}


# ---------------------------------------------------------------------------------------------------------------
# in-step kernel durations: rocprofv3 --kernel-trace --stats of this command, made by this invocation
# ---------------------------------------------------------------------------------------------------------------
MARK = "k_launch_floor"        # a kernel no train step contains: the child brackets its timed replays with it


def _replay_stats(trace_csv):
    """{kernel name: (calls, average us, total us)} of the dispatches BETWEEN the child's two marker launches — the
    timed replays only: the eager warm-up steps, the capture and the model set-up lie outside the bracket."""
    with open(trace_csv) as fh:
        rows = list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if MARK in r["Kernel_Name"]]
    if len(marks) < 2:
        return None, None
    body = rows[marks[-2] + 1:marks[-1]]
    agg = {}
    for r in body:
        a = agg.setdefault(r["Kernel_Name"], [0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span_us = (int(body[-1]["End_Timestamp"]) - int(body[0]["Start_Timestamp"])) / 1e3 if body else 0.0
    return {n: (c, t / c, t) for n, (c, t) in agg.items()}, span_us


def instep_profile(workload, bf16, steps=12, timeout=420):
    """Run ``bench.py --inner-profile`` (the same train step, nothing else) under ``rocprofv3 --kernel-trace --stats`` in
    a CHILD process and return ({kernel name: (calls, average us, total us)} over the child's timed REPLAYS only, number
    of replays, path of the profiler's stats CSV, span of the replays in us) — or None when the profiler is not
    available.  The profiler goes around the python program itself (no env/bash hop behind ``--``)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    keep = os.environ.get("IGCN_BENCH_PROFILE_DIR")
    out = keep or tempfile.mkdtemp(prefix="igcn_prof_")
    os.makedirs(out, exist_ok=True)
    cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--inner-profile", "--workload", workload, "--steps", str(steps),
           "--warmup", "3", "--bf16", "1" if bf16 else "0"]
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
        files = glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)
        traces = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files or not traces:
            print(f"[bench] rocprofv3 child failed (rc={r.returncode}): {r.stderr[-400:]}", file=sys.stderr)
            return None
        stats, span = _replay_stats(traces[0])
        if stats is None:
            print("[bench] rocprofv3 child: marker launches not found in the kernel trace", file=sys.stderr)
            return None
        if keep:
            shutil.copy(files[0], os.path.join(out, f"{workload}_kernel_stats.csv"))      # the profiler's own summary
            with open(os.path.join(out, f"{workload}_replay_kernel_stats.csv"), "w", newline="") as fh:
                w = csv.writer(fh)                        # the same columns over the timed replays only
                w.writerow(("Name", "Calls", "TotalDurationNs", "AverageNs", "CallsPerReplay", "UsPerReplay"))
                for n, (c, avg, tot) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
                    w.writerow((n, c, int(tot * 1e3), int(avg * 1e3), round(c / steps, 2), round(tot / steps, 2)))
                w.writerow(("SPAN first start -> last end per replay", "", int(span * 1e3), "", "", round(span / steps, 2)))
        return stats, steps, files[0], span
    except Exception as exc:                       # noqa: BLE001 — the profile is evidence, not the metric
        print(f"[bench] rocprofv3 child failed: {type(exc).__name__}: {exc}", file=sys.stderr)
        return None
    finally:
        if not keep:
            shutil.rmtree(out, ignore_errors=True)


PMC_JSON = next((p for p in (os.path.join(ROOT, "profiles", r + "_pmc", "traffic.json") for r in ("r05", "r04", "r03"))
                 if os.path.exists(p)), os.path.join(ROOT, "profiles", "r04_pmc", "traffic.json"))


def _attach_traffic(res):
    """``traffic``: HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950 correction, calibrated
    on a 256 MiB float4 copy; WRITE_SIZE x1).  PMC collection needs its own profiler passes, so the figure comes from
    the COMMITTED passes of tools/roofline_kernel.py (profiles/r05_pmc/, or $IGCN_BENCH_PMC_JSON) — a separate run of
    the same kernel on the same launch shape, not this invocation — and says so."""
    path = os.environ.get("IGCN_BENCH_PMC_JSON", PMC_JSON)
    try:
        with open(path) as fh:
            table = json.load(fh)
    except OSError:
        return
    for key, row in table.items():
        if key != "calibration" and res["kernel"].split("<")[0] == key and row.get("alg_bytes") == res["alg_bytes_per_launch"]:
            res["traffic"] = row["traffic_bytes"]
            res["traffic_source"] = (f"{os.path.relpath(path, ROOT)}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                     f"of tools/roofline_kernel.py on this launch shape ({row['us_median']} us median there)")
            res["frac_traffic"] = round(row["traffic_bytes"] / (res["us_per_launch"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)


def _pick(stats, prefix):
    """(name, calls, avg us, total us) of the kernels whose demangled name starts with ``prefix`` (largest total)."""
    best = None
    for name, (calls, avg, tot) in (stats or {}).items():
        n = name[5:] if name.startswith("void ") else name
        if n.startswith(prefix) and (best is None or tot > best[3]):
            best = (n.split("(")[0], calls, avg, tot)
    return best


# ---------------------------------------------------------------------------------------------------------------
# the scatter-aggregate kernel: replay timings with HIP events on the launch stream
# ---------------------------------------------------------------------------------------------------------------
def _propagate_args(plan, coef, h, bias, out, f, npg):
    n = h.shape[0]
    return (n, plan.n_edges, f, npg, h.data_ptr(), f, coef[2].data_ptr(), coef[1].data_ptr(), bias.data_ptr(),
            plan.tgt_ptr.data_ptr(), out.data_ptr(), f, 1)


def _time_graph(launch_all, reps=3):
    """Average device time (us) per replay of a hipGraph made of ``launch_all()``, between two HIP events on the
    launch stream (the current torch stream: igcn kernels are launched on it)."""
    launch_all()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):      # a process group's threads may be alive
        launch_all()
    g.replay()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3
        best = t if best is None else min(best, t)
    return best


def _propagate_timings(plan, ew, n, f, device, npg, hot_iters):
    """(hot us per launch, cold us per launch, sets used for the cold rotation)."""
    from igcn_amd import ops
    from igcn_amd._lib import call, stream_ptr
    coef = ops.GcnNorm.apply(ew, plan)
    bias = torch.zeros(f, device=device)
    h = torch.randn(n, f, device=device)
    out = torch.empty_like(h)
    args = _propagate_args(plan, coef, h, bias, out, f, npg)
    hot = _time_graph(lambda: [call("igcn_gcn_propagate_fwd", *args, stream_ptr()) for _ in range(hot_iters)]) / hot_iters
    # cold: every launch reads its own copy of the record stream and of h and writes its own out; the copies add up
    # to more than the 256 MiB Infinity Cache, so a buffer is evicted before its turn comes again
    per_set = coef[2].numel() * 4 + 2 * h.numel() * 4
    sets = max(3, int(1.5 * INFINITY_CACHE_BYTES / per_set) + 1)
    rec = [coef[2].clone() for _ in range(sets)]
    hs = [torch.randn(n, f, device=device) for _ in range(sets)]
    outs = [torch.empty(n, f, device=device) for _ in range(sets)]
    argl = []
    for k in range(sets):
        c = (coef[0], coef[1], rec[k], coef[3])
        argl.append(_propagate_args(plan, c, hs[k], bias, outs[k], f, npg))
    cold = _time_graph(lambda: [call("igcn_gcn_propagate_fwd", *a, stream_ptr()) for a in argl]) / sets
    return hot, cold, sets


def _floor_timings(n, f, device, iters, dense=False):
    """Launch floor of the scatter-aggregate grid: an EMPTY kernel of the same grid and one that only WRITES the
    output rows (igcn_launch_floor), per launch, hot replay."""
    from igcn_amd._lib import call, stream_ptr
    out = torch.empty(n, f, device=device)
    res = {}
    for mode, key in ((0, "empty_grid_us"), (1, "write_only_us")):
        t = _time_graph(lambda: [call("igcn_launch_floor", n, f, int(dense), mode, out.data_ptr(), stream_ptr())
                                 for _ in range(iters)]) / iters
        res[key] = round(t, 3)
    return res


def fused_stack_roofline(model, data, device, wl, stats):
    """Roofline of the kernel that carries the scatter-aggregate in the default step: the LDS-resident SGCN stack
    (igcn_sgcn_stack_fwd: gcn_norm + L x (X W^T, scatter-aggregate, bias, ReLU) + concatenation, one workgroup per
    graph).  Algorithmic bytes = SURVEY §8d "fused SGCN forward lower bound": 4 R H0 + 20 E + 4 R D per graph."""
    import ctypes
    from igcn_amd import ops
    from igcn_amd._lib import call, ptr, stream_ptr
    rois = wl["rois"]
    plan = ops.plan_for(data).replicate(2)
    convs = [model.conv1, *model.convs]
    f, layers, h0 = convs[0].out_channels, len(convs), data.x.shape[1]
    if not ops.sgcn_stack_supported(plan, rois, h0, f, layers):
        return None
    n, e = 2 * data.x.shape[0], plan.n_edges
    g = n // rois
    d = layers * f
    alg_bytes = g * (4 * rois * h0 + 20 * (e // g) + 4 * rois * d)
    # what the kernel itself moves: x, four 4-byte arrays per edge (src32, dst32, tgt_perm, weight), two per node
    # (tgt_ptr, loop_edge), the concatenated rows out
    kern_bytes = 4 * n * h0 + 16 * e + 8 * n + 4 * n * d
    emax = plan._stack_dims[1]
    ws = [c.lin.weight.detach() for c in convs]
    bs = [c.bias.detach() for c in convs]
    wp = (ctypes.c_void_p * layers)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * layers)(*[b.data_ptr() for b in bs])
    ew = torch.cat([data.edge_attr, data.edge_attr]).detach()
    x = torch.cat([data.x.detach(), data.x.detach()])

    def launcher(xb, ewb, arrs, out):
        return lambda: call("igcn_sgcn_stack_fwd", g, rois, emax, h0, f, layers, ptr(xb), ptr(ewb), ptr(arrs[0]),
                            ptr(arrs[1]), ptr(arrs[2]), ptr(arrs[3]), ptr(arrs[4]), wp, bp, ptr(out), None, stream_ptr())
    base = (plan.src32, plan.dst32, plan.tgt_ptr, plan.tgt_perm, plan.loop_edge)
    out = torch.empty(n, d, device=device)
    one = launcher(x, ew, base, out)
    hot_iters = 100
    hot = _time_graph(lambda: [one() for _ in range(hot_iters)]) / hot_iters
    sets = int(1.5 * INFINITY_CACHE_BYTES / kern_bytes) + 1
    fns = [launcher(x.clone(), ew.clone(), tuple(a.clone() for a in base), torch.empty(n, d, device=device))
           for _ in range(sets)]
    cold = _time_graph(lambda: [fn() for fn in fns]) / sets
    picked = _pick(stats[0], "k_sgcn_stack_fwd") if stats else None
    front = _pick(stats[0], "k_sgcn_front_fwd") if stats else None
    if front is not None:
        # round 5: the default step runs the stack inside the FRONT kernel of the image branch (igcn_sgcn_front_fwd: the
        # per-graph plan build, the masks of both passes, loss_probability, the SNP mask and the stack in one launch,
        # carrying the step's dropout rider).  `achieved` keeps the contract's definition — SURVEY 8d's fused lower bound
        # per graph-pass x the 512 graph-passes of the launch / the launch's duration — and so charges the whole launch
        # to the stack's 9.2 MB; `kernel_bytes_per_launch` is what the launch itself reads and writes (int64 edge list,
        # x, weights in; both plans' seven arrays, x_in, ew_in, e, xcat out), the dropout factors its rider writes not
        # counted.  The stack alone (the kernel the other routes launch) keeps its cold / hot replay beside it.
        gr, eg = g // 2, e // g
        front_bytes = gr * (16 * eg + 4 * eg + 4 * rois * h0) * 2 + 4 * n * d + 4 * n * h0 + 4 * e + 2 * e \
            + 3 * (16 * (e // 2) + 12 * (n // 2))
        picked = front
        kern_bytes_front = front_bytes
    us = picked[2] if picked else cold
    src = ("rocprofv3 --kernel-trace of this command (child process): average over the launches of its timed replays"
           if picked else ("HIP events, cold replay (the rocprofv3 child of this run failed)" if stats is None else
                           "HIP events, cold replay (the step does not launch this kernel)"))
    gbs = alg_bytes / (us * 1e-6) / 1e9
    res = {"bound": "hbm", "kernel": picked[0] if picked else "k_sgcn_stack_fwd", "achieved": round(gbs, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
            "alg_bytes_per_launch": alg_bytes, "us_per_launch": round(us, 3), "timing": src,
            "launches_profiled": picked[1] if picked else 0, "kernel_bytes_per_launch": kern_bytes,
            "frac_kernel_bytes": round(kern_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "replay_hot_us": round(hot, 3), "frac_replay_hot": round(alg_bytes / (hot * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "replay_cold_us": round(cold, 3),
            "frac_replay_cold": round(alg_bytes / (cold * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "cold_rotation": f"{sets} buffer sets = {sets * kern_bytes / 2**20:.0f} MiB > 256 MiB Infinity Cache",
            "launch": f"{g} graphs x {e // g} edges (both passes of a step), H0={h0}, F={f}, L={layers}: "
                      "ONE launch for gcn_norm + every GCNConv + ReLU + concatenation",
            "note": "latency-bound by construction: 512 workgroups (2 per CU) each run ~12 barrier-separated LDS phases; "
                    "the kernel replaces 7 launches (norm x2, GEMM + scatter-aggregate per layer, concat), whose "
                    "own scatter-aggregate launch is roofline_scatter_standalone"}
    if front is not None:
        res["kernel_bytes_per_launch"] = kern_bytes_front
        res["frac_kernel_bytes"] = round(kern_bytes_front / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        res["launch"] = (f"{g} graph-passes x {e // g} edges: per-graph plan build + masks + loss_probability + SNP mask + "
                         f"gcn_norm + {layers} x GCNConv + ReLU + concatenation, and the step's dropout rider, in ONE launch")
        res["note"] = ("latency-bound by construction (one workgroup per graph-pass, ~20 barrier-separated LDS phases: "
                       "tools/front_probe.py); replaces plan build + mask + stack launches (11.7 + 6.3 + 9.0 us); the "
                       "stack kernel alone: replay_hot_us / replay_cold_us")
    _attach_traffic(res)
    return res


def scatter_roofline(data, device, wl, stats, hot_iters=200):
    """Roofline of igcn_gcn_propagate_fwd (F=16) exactly as the train step launches it: both passes batched = 2
    copies of the batch's graphs.  SURVEY §8d algorithmic bytes = (20*E' + 8*R*F) per graph (int64 endpoints + fp32
    coefficient per edge, each feature row read and written once)."""
    from igcn_amd import ops
    rois = wl["rois"]
    plan = ops.plan_for(data).replicate(2)
    n, f = 2 * data.x.shape[0], HIDDEN
    n_graphs = n // rois
    e_prime = plan.n_edges // n_graphs                       # GDC / dense graphs store their self-loops: E' = E
    alg_bytes = n_graphs * (20 * e_prime + 8 * rois * f)
    # what the kernel itself moves: one 8-byte (neighbour, coefficient) record per edge, the row pointers and the
    # self-loop coefficient per node, each feature row in and out once
    kern_bytes = n_graphs * (8 * e_prime + 4 * (rois + 1) + 4 * rois + 8 * rois * f)
    dense = plan.n_edges >= 16 * n
    hot, cold, sets = _propagate_timings(plan, torch.cat([data.edge_attr, data.edge_attr]), n, f, device,
                                         rois if dense else 0, hot_iters if not dense else 20)
    prefix = "k_gcn_propagate_fwd"
    picked = _pick(stats[0], prefix) if stats else None
    us = picked[2] if picked else cold
    src = ("rocprofv3 --kernel-trace of this command (child process): average over the launches of its timed replays"
           if picked else ("HIP events, cold replay (the rocprofv3 child of this run failed)" if stats is None else
                           "HIP events, cold replay (the step does not launch this kernel)"))
    gbs = alg_bytes / (us * 1e-6) / 1e9
    fallback_name = "k_gcn_propagate_fwd_lds<4>" if dense else "k_gcn_propagate_fwd_q<4>"
    res = {"bound": "hbm", "kernel": picked[0] if picked else fallback_name, "achieved": round(gbs, 1),
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
           "alg_bytes_per_launch": alg_bytes, "us_per_launch": round(us, 3), "timing": src,
           "launches_profiled": picked[1] if picked else 0,
           "kernel_bytes_per_launch": kern_bytes,
           "frac_kernel_bytes": round(kern_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
           "replay_hot_us": round(hot, 3), "frac_replay_hot": round(alg_bytes / (hot * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
           "replay_cold_us": round(cold, 3),
           "frac_replay_cold": round(alg_bytes / (cold * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
           "cold_rotation": f"{sets} buffer sets = {sets * (kern_bytes) / 2**20:.0f} MiB > 256 MiB Infinity Cache",
           "launch": f"{n_graphs} graphs x {e_prime} edges (both passes of a step), F={f}"}
    if dense:
        res["note"] = ("SURVEY 8d counts int64 endpoint pairs (20 B per edge); this kernel streams the plan's 8-byte "
                       "(neighbour, coefficient) records, so `frac` over-states it — frac_kernel_bytes / frac_traffic "
                       "(what the kernel really moves) are the figures to read at this shape")
    try:
        res["floor"] = _floor_timings(n, f, device, hot_iters if not dense else 50, dense)
        res["floor"]["note"] = ("same grid, hot replay: an empty kernel and one that only writes the output rows; "
                                "frac_ceiling = alg bytes / write-only time / peak")
        res["floor"]["frac_ceiling"] = round(alg_bytes / (res["floor"]["write_only_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    except Exception as exc:                       # noqa: BLE001
        res["floor"] = {"error": f"{type(exc).__name__}: {exc}"}
    _attach_traffic(res)
    return res


def dense_roofline(data, wl, stats):
    """configs[4] on the dense-block path: the scatter-aggregate of a GCNConv layer is k_ds_agg (csrc/sgcn_dense.hip) —
    ONE launch per layer for BOTH passes of the step, reading edge_attr as the dense matrix it is (4 bytes per edge, once)
    and recomputing mask and coefficient on the fly.  ``frac`` follows the contract (SURVEY §8d algorithmic bytes: int64
    endpoint pair + fp32 coefficient per edge and pass, 20 E' + 8 R F per graph and pass — none of which this kernel reads,
    so the figure exceeds 1); the figures to read are ``frac_kernel_bytes`` (what the kernel moves) and ``frac_traffic``
    (PMC).  Duration: average over the in-step launches of the rocprofv3 child of this command."""
    picked = _pick(stats[0], "k_ds_agg<") if stats else None      # not k_ds_aggT (the transposed pass)
    if picked is None:
        return None
    rois, f = wl["rois"], HIDDEN
    g = data.x.shape[0] // rois
    e = rois * rois
    copies = 2
    alg_bytes = copies * g * (20 * e + 8 * rois * f)
    # ew once (both passes), h' rows in, AGG + activations + the next layer's h' out, per-node factors
    kern_bytes = 4 * g * e + copies * g * rois * f * 4 * 4 + 4 * g * rois * 4
    us = picked[2]
    res = {"bound": "hbm", "kernel": picked[0], "achieved": round(alg_bytes / (us * 1e-6) / 1e9, 1),
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
           "traffic": None, "alg_bytes_per_launch": alg_bytes, "us_per_launch": round(us, 3),
           "timing": "rocprofv3 --kernel-trace of this command (child process): average over the launches of its timed replays",
           "launches_profiled": picked[1], "kernel_bytes_per_launch": kern_bytes,
           "frac_kernel_bytes": round(kern_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
           "launch": f"{g} complete graphs x {e} edges, BOTH passes of the step in one launch, F={f}",
           "note": "dense-block path: no index arrays are read (edge_index is verified once per step by workgroups that ride "
                   "in k_ds_deg's launch), so "
                   "the contract's `frac` (int64 pairs + coefficient per edge and pass) exceeds 1; frac_kernel_bytes / "
                   "frac_traffic are the honest figures.  The launch also carries 0.54 GFLOP of fp32 MFMA (3.4 us at "
                   "the 157 TFLOP/s peak)."}
    others = {}
    check_rides = _pick(stats[0], "k_ds_check") is None     # no launch of its own: its workgroups ride in k_ds_deg's
    for name in ("k_ds_aggT", "k_ds_mask_bwd", "k_ds_deg", "k_ds_check"):
        pk = _pick(stats[0], name + ("<" if name != "k_ds_check" else ""))
        if pk:
            byt = 16 * g * e if name == "k_ds_check" else 4 * g * e
            if name == "k_ds_deg" and check_rides:
                byt += 16 * g * e                            # + the int64 index pairs the riding check streams
            others[name] = {"us": round(pk[2], 2), "launches_profiled": pk[1], "edge_bytes_per_launch": byt,
                            "frac_edge_bytes": round(byt / (pk[2] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
            if name == "k_ds_deg" and check_rides:
                others[name]["note"] = ("4 B weights + 16 B index pairs per edge: the batch's structure check rides in "
                                        "this launch (k_ds_check as a launch of its own: 27.6 us + 12.5 us)")
    # PMC traffic of the other passes, where the committed passes cover them at this launch shape (32 graphs x R = 512)
    try:
        with open(os.environ.get("IGCN_BENCH_PMC_JSON", PMC_JSON)) as fh:
            table = json.load(fh)
    except OSError:
        table = {}
    for name, row in others.items():
        t = table.get(name)
        if name == "k_ds_deg" and check_rides:
            t = None                                         # (the committed passes measured the weights-only launch)
        if t and g == 32 and e == 512 * 512:
            row["traffic"] = t["traffic_bytes"]
            row["frac_traffic"] = round(t["traffic_bytes"] / (row["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    res["other_edge_passes"] = others
    _attach_traffic(res)
    return res


def mfma_roofline(model, wl, device, stats, bf16):
    """Dense feature transforms against the matrix-core peak of their operand type: lin1 (the widest product of the
    heads) and the key|value projection of the cross-attention — hot replay, HIP events on the launch stream — plus the
    in-step total of the GEMM kernels from the rocprofv3 child."""
    from igcn_amd import ops
    kind = "bf16" if bf16 else "f32"
    peak = MFMA_PEAK_TFLOPS[kind]
    out = {"bound": "mfma", "operands": kind, "peak": peak, "unit": "TFLOP/s", "products": []}
    g = 2 * wl["graphs"]
    d = LAYERS * HIDDEN
    n_top = sum(wl["pool"][2:])
    iters = 50

    def add(name, us, flops, byt, dims, kernel):
        tf = flops / (us * 1e-6) / 1e12
        out["products"].append(dict(name=name, **dims, us=round(us, 2), achieved=round(tf, 2), frac=round(tf / peak, 4),
                                    frac_hbm=round(byt / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), kernel=kernel))

    m, n, k = g, 64, wl["rois"] * d + 32                     # lin1 (:299): the widest product of the heads
    a, b, bias = torch.randn(m, k, device=device), torch.randn(n, k, device=device), torch.zeros(n, device=device)
    us = _time_graph(lambda: [ops.gemm_nt(a, b, bias, 1, bf16=bf16) for _ in range(iters)]) / iters
    add("lin1", us, 2.0 * m * n * k, 4.0 * (m * k + n * k + m * n), dict(M=m, N=n, K=k), "k_gemm_bf16" if bf16 else "k_gemm_f32")
    # the packed in-projection of the cross-attention (:240) as the model runs it: queries [g * rois, d] -> d and
    # key | value [g * n_top, d] -> 2 d in one launch (fp32: the streaming kernel igcn_proj_fwd_pair; bf16: grouped GEMM)
    mq, mk = g * wl["rois"], g * n_top
    q2, m2 = torch.randn(mq, d, device=device), torch.randn(mk, d, device=device)
    w, pb = torch.randn(3 * d, d, device=device), torch.zeros(3 * d, device=device)
    us = _time_graph(lambda: [ops._proj_forward(q2, m2, w, pb, d, bf16) for _ in range(iters)]) / iters
    add("in_proj", us, 2.0 * d * d * (mq + 2 * mk), 4.0 * (2 * mq * d + 3 * mk * d + 3 * d * d),
        dict(M=mq + mk, N=2 * d, K=d), "k_gemm_bf16_grouped" if bf16 else "k_proj_fwd")
    best = max(out["products"], key=lambda p: p["frac"])
    out["achieved"], out["frac"], out["kernel"] = best["achieved"], best["frac"], f"{best['kernel']} ({best['name']})"
    # the cross-attention core as the step runs it (in-step durations of this run's profiled child) with the MATRIX-PIPE
    # busy fraction of the same kernels from the committed counter pass (profiles/r05_pmc/mfma*.json: SQ_VALU_MFMA_BUSY_CYCLES
    # / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), tools/mfma_run.sh — a separate run, labelled as such)
    counters = {}
    for fn in ("mfma.json", "mfma_exact_fp32.json"):
        try:
            with open(os.path.join(ROOT, "profiles", "r05_pmc", fn)) as fh:
                for key, row in json.load(fh)["kernels"].items():
                    counters.setdefault(key.split(" #")[0].split("<")[0], row)
        except (OSError, KeyError, ValueError):
            pass
    if stats and not bf16:
        lq, lk, heads = wl["rois"], n_top, 2
        for pref, mult, what in (("k_attn_split_fwd", 2, "attention forward (split bf16 operands)"),
                                 ("k_attn_split_bwd", 5, "attention backward (split bf16 operands)"),
                                 ("k_attn_mfma_fwd", 2, "attention forward (exact fp32)"),
                                 ("k_attn_mfma_bwd", 5, "attention backward (exact fp32)")):
            hit = _pick(stats[0], pref)
            if hit is None:
                continue
            flops = 2.0 * mult * g * heads * lq * lk * (d // heads)
            row = dict(name=what, kernel=hit[0].split("(")[0], us=round(hit[2], 2), useful_tflops=round(flops / hit[2] / 1e6, 1),
                       note="useful flops 2 x %d x B H Lq Lk hd; the split core issues 2-3 bf16 instructions per product" % mult
                       if "split" in pref else "useful flops 2 x %d x B H Lq Lk hd" % mult)
            c = counters.get(pref)
            if c:
                row["mfma_busy_frac"] = c["mfma_busy_frac"]
                row["mfma_busy_source"] = "profiles/r05_pmc: separate counter pass, %s us there" % c["us_median_unprofiled_pass"]
            out.setdefault("attention_core", []).append(row)
    if stats:
        tot = sum(t for nme, (c, a_, t) in stats[0].items()
                  if ("k_gemm_" in nme or "k_proj_" in nme or "k_head_bwd" in nme) and "reduce" not in nme)
        red = sum(t for nme, (c, a_, t) in stats[0].items() if "reduce" in nme)
        out["instep_gemm_us_per_step"] = round(tot / stats[1], 1)
        out["instep_reduction_kernels_us_per_step"] = round(red / stats[1], 1)
    return out


def _pick_cores(usable, pci=None):
    """``usable`` host CPUs to pin the baseline to: distinct PHYSICAL cores (one hardware thread each), taken from the
    CPUs local to the GPU (``/sys/bus/pci/devices/<bdf>/local_cpulist``: the boxes of one host then land on different
    NUMA nodes instead of all on cpu0-15) and inside the process's affinity mask; falls back to the mask's own order."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None

    def parse(text):
        out = []
        for part in text.strip().split(","):
            if "-" in part:
                a, b = part.split("-")
                out += list(range(int(a), int(b) + 1))
            elif part:
                out.append(int(part))
        return out
    # how busy every logical CPU is right now (two /proc/stat samples 0.3 s apart): a GPU box is a share of a host whose
    # other tenants may sit on the very cores a fixed choice would take (first run of this round: load average 22 on the 16
    # GPU-local cores, 2.6x between the fastest and the slowest of five steps)
    def busy_now():
        def snap():
            out = {}
            try:
                with open("/proc/stat") as fh:
                    for ln in fh:
                        if ln.startswith("cpu") and ln[3].isdigit():
                            f = ln.split()
                            v = [int(x) for x in f[1:9]]
                            out[int(f[0][3:])] = (sum(v), v[3] + v[4])
            except (OSError, ValueError, IndexError):
                return {}
            return out
        a = snap()
        time.sleep(0.3)
        b = snap()
        return {c: 1.0 - (b[c][1] - a[c][1]) / max(1, b[c][0] - a[c][0]) for c in a if c in b}
    busy = busy_now()
    local = None
    if pci:
        try:
            with open(f"/sys/bus/pci/devices/{pci}/local_cpulist") as fh:
                local = [c for c in parse(fh.read()) if c in set(allowed)]
        except (OSError, ValueError):
            local = None
    order = (local or []) + [c for c in allowed if c not in set(local or [])]
    seen, cand = set(), []
    for rank, c in enumerate(order):
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as fh:
                key = tuple(sorted(parse(fh.read())))
        except (OSError, ValueError):
            key = (c,)
        if key in seen:
            continue
        seen.add(key)
        load = max(busy.get(t, 0.0) for t in key)              # a physical core is as busy as its busiest thread
        cand.append((load > 0.25, round(load, 1), rank, c))      # idle cores first (GPU-local ones before the others)
    cand.sort()
    cores = sorted(c for _, _, _, c in cand[:usable])
    return cores or None


def cpu_baseline_child(workload, seconds, cores):
    """Runs in a fresh process that never touches the GPU (``bench.py --cpu-baseline-child``): pinned to ``cores``
    (one thread per core, OMP_PROC_BIND / OMP_PLACES set by the parent before this interpreter started), it times the
    oracle's train step — the reference's step, kernel/train_eval_sgcn_img_snps.py:511-548 — at the headline's own
    batch size and prints one JSON object."""
    from types import SimpleNamespace
    from oracle import go_network as OG, sgcn_img_snp as OS
    wl = WORKLOADS[workload]
    if cores:
        try:
            os.sched_setaffinity(0, set(cores))
        except (AttributeError, OSError):
            cores = None
    threads = len(cores) if cores else _usable_cores()
    torch.set_num_threads(threads)
    go_snps, adj, pool_dim = synth.go_hierarchy(wl["pool"], seed=0)
    pool, rois = wl["pool"], wl["rois"]
    a_g, a = synth.go_sparse_inputs(go_snps, adj)
    idx = OG.go_index_sets(a_g, a, list(pool), 2)
    shapes = dict(OS.sgcn_param_shapes(LAYERS, HIDDEN, rois=rois))
    shapes.update({"go_network." + k: v for k, v in OG.go_param_shapes(idx, l_dim=32, d_att=LAYERS * HIDDEN).items()})
    gen = torch.Generator().manual_seed(0)
    sd = {}
    for k, s in shapes.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_var") or (len(s) == 1 and k.endswith(".weight")):
            sd[k] = torch.ones(s)
        elif k.endswith("running_mean"):
            sd[k] = torch.zeros(s)
        elif ".t." in k or ".t_D." in k:
            sd[k] = 1 + 0.1 * torch.randn(s, generator=gen)
        else:
            sd[k] = (torch.rand(s, generator=gen) * 2 - 1) / max(1.0, float(s[-1] if len(s) > 1 else s[0])) ** 0.5
    sd = OS.make_leaf_state(sd)
    cfg = SimpleNamespace(num_layers=LAYERS, rois=rois, image_only=False, rbf_gamma=0.01)
    # ALWAYS the headline's batch size (B = 256 at configs[2]; the dense 512-ROI workload is 1.3 graphs/s on a host
    # share, so configs[4] samples B = 4), one untimed warm-up step, then >= 5 timed steps or until the bound is used up
    b_full = wl["graphs"]
    b = b_full if not wl["dense"] else 4
    data = Batch.from_data_list(synth.brain_graph_list(b, seed=1000, rois=rois, tsne_dim=90, dense=wl["dense"]))
    opt, times = None, []
    load0 = os.getloadavg()[0] if hasattr(os, "getloadavg") else None
    t_end = time.perf_counter() + seconds
    _, _, opt = OS.train_step(sd, cfg, idx, data, lr=1e-3, dropout=True, faithful=True, opt=opt)
    while len(times) < 6 or (time.perf_counter() < t_end and len(times) < 12):
        t0 = time.perf_counter()
        _, _, opt = OS.train_step(sd, cfg, idx, data, lr=1e-3, dropout=True, faithful=True, opt=opt)
        times.append(time.perf_counter() - t0)
    ts = sorted(times)
    med = ts[len(ts) // 2]
    # `value` = the FASTEST timed step: the host is shared (load average 20-60 on a 256-thread box, 16 usable cores), a step
    # is disturbed or it is not, and the minimum time is the estimator other tenants move least — two runs of this round
    # gave medians of 120.3 and 80.2 graphs/s but fastest steps of 129.6 and 125.4 (3 %).  min / median / max stay in the line.
    print(json.dumps({
        "value": round(b / ts[0], 2), "unit": "graphs/s", "cores": threads, "kind": "port", "estimator": "fastest timed step",
        "graphs_per_s_min_median_max": [round(b / ts[-1], 2), round(b / med, 2), round(b / ts[0], 2)],
        "spread": round((ts[-1] - ts[0]) / med, 3), "steps_timed": len(ts), "batch": b,
        "pinned_cpu_ids": cores, "host_load_1min_at_start": load0,
        "host_logical_cpus": os.cpu_count(), "physical_cores": _physical_cores(), "usable_cores": _usable_cores(),
        "torch": torch.__version__,
        "sample": f"{len(ts)} timed train steps (+1 warm-up) of B={b} graphs (the headline runs B={b_full}; same model / "
                  f"GO DAG, fp32), fastest step; oracle faithful mode; {threads} threads pinned one per physical core"}))


def cpu_baseline(workload, seconds=20.0, device=None):
    """The oracle (CPU restatement of the reference's step, faithful mode: per-sample sparse loop) timed on this box's
    host cores, in a CHILD process pinned to as many distinct physical cores as the cgroup grants (local to the GPU's
    NUMA node when the PCI address is known) — VERDICT r4 #8: the unpinned 20-second sample moved 2.7x between boxes."""
    usable = _usable_cores()
    pci = None
    try:
        pr = torch.cuda.get_device_properties(device if device is not None else 0)
        pci = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:                              # noqa: BLE001
        pci = None
    cores = _pick_cores(usable, pci)
    env = dict(os.environ)
    n = len(cores) if cores else usable
    env.update({"OMP_NUM_THREADS": str(n), "MKL_NUM_THREADS": str(n), "OMP_PROC_BIND": "close", "OMP_PLACES": "cores",
                "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--workload", workload,
           "--cpu-baseline-seconds", str(seconds), "--cpu-baseline-cores", ",".join(str(c) for c in cores or [])]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=max(180.0, 8 * seconds))
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"value": None, "unit": "graphs/s", "cores": n, "kind": "port",
                    "error": (r.stderr or r.stdout)[-400:]}
        out = json.loads(line[-1])
        out["gpu_pci"] = pci
        return out
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "graphs/s", "cores": n, "kind": "port", "error": "baseline child timed out"}


def _usable_cores():
    """Host cores this process may actually run on: the smaller of the affinity mask and the cgroup CPU quota (a GPU
    box shows every CPU of the host but grants a share of them: 256 OpenMP threads on a 16-core share thrash)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    per = int(fh.read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False)
    except Exception:                              # noqa: BLE001
        return None


def pipeline_bench(gstep, wl, device, steps, warmup, resident_ms):
    """Loader-fed throughput (the reference's step starts at ``for data in loader: data.to(device)``,
    kernel/train_eval_sgcn_img_snps.py:515-517): NEW graphs every step, three ways, each ending in
    GraphedTrainStep.load (device-to-device copies into the captured step's static inputs) + one replay.
      host        dataset on the host: a producer thread collates (vectorised, bit-identical to Batch.from_data_list)
                  into pinned memory and uploads on a copy stream (igcn_amd.loader.HostFeeder)
      device      dataset resident in HBM: the collation is a few device gathers (DeviceFeeder)
      device_gdc  dense connectivity matrices resident in HBM: GDC pre-transform (PPR inverse, top-k, normalise) +
                  collation of the batch on the device every step (igcn_gdc_topk), then load + replay
    The reference's own collation loop (igcn_amd.data.Batch.from_data_list restates it) is timed beside them."""
    from igcn_amd.data import Data
    from igcn_amd.gdc import batch_from_dense
    from igcn_amd.loader import DeviceFeeder, HostFeeder, UniformGraphStore
    b = wl["graphs"]
    subjects = 4 * b
    graphs = synth.brain_graph_list(subjects, seed=3000, rois=wl["rois"], tsne_dim=90, dense=wl["dense"])
    adj_all = torch.stack([g.A for g in graphs])
    keep = ("x", "edge_index", "edge_attr", "snps_feat", "y", "clini_score", "tsne_fdim", "clust_y")
    slim = [Data(**{k: getattr(g, k) for k in keep}) for g in graphs]
    out = {"subjects": subjects, "graphs_per_step": b, "steps": steps, "blocks": 3, "reported": "median block",
           "host_logical_cpus": os.cpu_count(),
           "usable_cores": _usable_cores(), "feeder_threads": 1,
           "resident_ms_per_step": resident_ms}
    t0 = time.perf_counter()
    for i in range(3):
        Batch.from_data_list(slim[i * b:(i + 1) * b])
    out["reference_collate_ms_per_batch"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
    cur = torch.cuda.current_stream()

    blocks = 3
    feed_steps = warmup + blocks * steps + 1             # what every feeder below is asked for

    def run(feed, consume):
        """THREE timed blocks of `steps` fed steps, the median reported (a host feeder shares the box's CPUs with
        whatever else runs there: one descheduled producer thread used to decide a single 60-step figure)."""
        it = iter(feed)
        for _ in range(warmup):
            consume(next(it))
        per = []
        for _ in range(blocks):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(steps):
                consume(next(it))
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t) / steps * 1e3)
        for _ in it:                                     # drain (the producer thread ends)
            pass
        ms = sorted(per)[len(per) // 2]
        return {"ms_per_step": round(ms, 3), "graphs_per_s": round(b / ms * 1e3, 1),
                "vs_resident": round(resident_ms / ms, 3), "ms_per_step_blocks": [round(v, 3) for v in per]}

    # host-fed
    host_store = UniformGraphStore(slim, "cpu", pin=True)
    t0 = time.perf_counter()
    slot = host_store.batch(torch.arange(b))
    for _ in range(5):
        host_store.batch(torch.randint(0, subjects, (b,)), out=slot)
    out["vectorised_collate_ms_per_batch"] = round((time.perf_counter() - t0) / 6 * 1e3, 3)

    def consume_host(batch):
        cur.wait_event(batch.ready)
        gstep.load(batch)
        batch.release()
        gstep()
    out["host"] = run(HostFeeder(host_store, b, device, feed_steps), consume_host)
    # device-resident dataset
    dev_store = UniformGraphStore(slim, device)

    def consume_dev(batch):                                  # hand-over by events: the batch was built on a side stream
        cur.wait_event(batch.ready)
        gstep.load(batch)
        batch.release()
        gstep()
    # the gather as ONE launch on the launch stream straight into the step's static inputs (no slot, no second queue) ...
    out["device"] = run(DeviceFeeder(dev_store, b, feed_steps, into=gstep.data), consume_dev)
    out["device"]["how"] = "igcn_gather_batch into the step's inputs, on the launch stream"
    # ... and one batch ahead on a side stream into a staging slot + hand-over copy (what device_gdc has to do)
    out["device_side_stream"] = run(DeviceFeeder(dev_store, b, feed_steps), consume_dev)
    # dense connectivity on the device -> GDC + collation of batch k + 1 on a second stream while step k replays
    if not wl["dense"]:
        from igcn_amd.loader import DeviceGdcFeeder
        adj_dev = adj_all.to(device)
        cols = {k: dev_store.cols[k] for k in ("x", "snps_feat", "y", "clini_score", "tsne_fdim", "clust_y")}
        first = batch_from_dense(adj_dev[:b], cols["x"][:b], top_k=3, alpha=0.05, check=True)
        if first.edge_index.shape == gstep.data.edge_index.shape:
            out["device_gdc"] = run(DeviceGdcFeeder(adj_dev, cols, b, feed_steps, top_k=3, alpha=0.05, seed=1),
                                    consume_dev)
            out["device_gdc"]["overlap"] = "GDC + collate of batch k+1 on a side stream during the replay of step k"
            gstep.plan.check()
            if os.environ.get("IGCN_BENCH_GDC_SERIAL", "0") == "1":      # A/B: the transform on the launch stream
                gen = torch.Generator(device=device).manual_seed(1)

                def gdc_feed():
                    for _ in range(feed_steps):
                        idx = torch.randint(0, subjects, (b,), generator=gen, device=device)
                        sel = {k: torch.index_select(v, 0, idx) for k, v in cols.items()}
                        yield batch_from_dense(torch.index_select(adj_dev, 0, idx), sel.pop("x"), top_k=3, alpha=0.05,
                                               check=False, snps_feat=sel["snps_feat"].reshape(b, -1),
                                               y=sel["y"].reshape(-1), clini_score=sel["clini_score"],
                                               tsne_fdim=sel["tsne_fdim"].reshape(b, -1),
                                               clust_y=sel["clust_y"].reshape(-1))
                def consume_serial(batch):
                    gstep.load(batch)
                    gstep()
                out["device_gdc_serial"] = run(gdc_feed(), consume_serial)
        else:
            out["device_gdc"] = {"skipped": "the GDC of these matrices does not keep top_k entries in every column"}
    return out


def step_traffic(workload, kernel_us_per_step):
    """The WHOLE step against the HBM roof: bytes per replay summed over every kernel of the step — the COMMITTED PMC
    passes of tools/step_traffic.sh (profiles/r05_step_traffic_<workload>.csv: --pmc FETCH_SIZE and --pmc WRITE_SIZE in
    runs of their own over marker-bracketed replays) — divided by THIS run's kernel time per replay."""
    path = next((p for p in (os.path.join(ROOT, "profiles", f"{r}_step_traffic_{workload}.csv") for r in ("r05", "r04"))
                 if os.path.exists(p)), os.path.join(ROOT, "profiles", f"r04_step_traffic_{workload}.csv"))
    try:
        import csv
        with open(path, newline="") as fh:
            tot = [r for r in csv.DictReader(fh) if r["kernel"] == "TOTAL"][0]
        nbytes = (float(tot["MB_read_per_replay"]) + float(tot["MB_written_per_replay"])) * 1e6
        rate = nbytes / kernel_us_per_step / 1e3                       # GB/s
        return {"bound": "hbm", "traffic": int(nbytes), "read_bytes": int(float(tot["MB_read_per_replay"]) * 1e6),
                "written_bytes": int(float(tot["MB_written_per_replay"]) * 1e6), "kernel_us_per_step": kernel_us_per_step,
                "achieved": round(rate, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(rate / 8000.0, 4),
                "launches_in_table": float(tot["launches_per_replay"]),
                "traffic_source": os.path.relpath(path, ROOT) + ": PMC bytes of every kernel of the replay (committed passes); "
                                  "time = this run's kernel time per replay"}
    except Exception:                                                  # noqa: BLE001 — a missing table is not an error
        return None


def stress_child(timeout=300):
    """BASELINE configs[4] beside the headline: a short ``--workload stress`` run of this script in a CHILD process
    (its own model, graph capture, in-step rocprofv3 profile and bounded CPU-oracle sample) whose result line is folded
    into the default line as ``stress``.  None (with the reason on stderr) when the child fails: the headline stands."""
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "stress", "--steps", "10", "--warmup", "3",
           "--cpu-baseline-seconds", "8", "--no-stress", "--no-pipeline", "--blocks", "3"]
    # the child profiles into its own directory (a kept parent directory would hand it the parent's kernel summary)
    env = dict(os.environ)
    keep = env.pop("IGCN_BENCH_PROFILE_DIR", None)
    if keep:
        env["IGCN_BENCH_PROFILE_DIR"] = os.path.join(keep, "stress_child")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        if r.returncode != 0 or not lines:
            print(f"[bench] stress child failed (rc={r.returncode}): {r.stderr[-600:]}", file=sys.stderr)
            return None
        full = json.loads(lines[-1])
    except Exception as exc:                       # noqa: BLE001
        print(f"[bench] stress child failed: {type(exc).__name__}: {exc}", file=sys.stderr)
        return None
    keep = ("value", "unit", "ms_per_step", "timing", "steps", "warmup", "dtype", "loss", "roofline", "roofline_mfma",
            "roofline_step", "cpu_baseline", "profile", "edge_pipeline")
    out = {k: full[k] for k in keep if k in full}
    out["workload"] = full["config"]["workload"]
    out["graphs_per_gpu"] = full["config"]["graphs_per_gpu"]
    return out


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher around it: start the N ranks as CHILD processes
    (``python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>``, rendezvous on 127.0.0.1), forward
    rank 0's JSON line and return the launcher's exit code.  Runs before this process has made any GPU call — never
    an exec of a GPU-initialised process; ``torch.cuda.device_count()`` does not initialise the runtime."""
    have = torch.cuda.device_count()
    if have < n and os.environ.get("IGCN_BENCH_ONE_DEVICE", "0") != "1":
        print(f"bench.py --gpus {n}: this box has {have} GPU(s).  One process per GPU needs {n}; for a control-flow "
              "rehearsal on one device set IGCN_BENCH_ONE_DEVICE=1 (ranks share cuda:0, gradients over gloo).",
              file=sys.stderr)
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in r.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)                       # anything else the ranks wrote to stdout
    if lines:
        print(lines[-1])
    elif r.returncode == 0:
        print("bench.py: the ranks exited cleanly but printed no result line", file=sys.stderr)
        return 4
    return r.returncode


def sweep(ns):
    """IGCN_BENCH_SWEEP="1,2,4,8": one invocation, result lines per N.  Every run is a fresh CHILD process of this script
    started before this process has made any GPU call (the child self-launches its ranks); the N = 1 value is handed to
    the larger runs, whose lines then carry ``weak_scaling_efficiency_vs_n1`` = value_N / (N * value_1).

    At N > 1 EVERY gradient-exchange form is run, one child each, so that a single hardware run says which to keep
    (VERDICT r4 #5): ``two_graphs`` = [zero_grad .. backward + pack] graph -> igcn_comm_allreduce -> [Adam] graph (the
    default), ``two_buckets`` = three graphs with the heads' all-reduce on a side stream beside the rest of the backward
    (IGCN_DP_TWO_BUCKETS=1), ``in_graph`` = one graph with the collective captured (IGCN_COMM_IN_GRAPH=1; the ranks agree on it through
    a MIN-reduced flag and fall back together).  The line of each child carries ``exchange_form``; a child that hangs is
    killed after IGCN_BENCH_SWEEP_TIMEOUT seconds (default 600) and only loses its own line."""
    env = dict(os.environ)
    env.pop("IGCN_BENCH_SWEEP")
    n1, rc = None, 0
    tmo = float(os.environ.get("IGCN_BENCH_SWEEP_TIMEOUT", "600"))
    for n in ns:
        argv = [a for a in sys.argv[1:]]
        if "--gpus" in argv:
            k = argv.index("--gpus")
            del argv[k:k + 2]
        argv = [a for a in argv if not a.startswith("--gpus=")]
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n), *argv]
        if n > 1:
            cmd += [f for f in ("--no-roofline", "--no-cpu-baseline", "--no-pipeline") if f not in argv]
        if n1 is not None:
            env["IGCN_BENCH_N1_VALUE"] = str(n1)
        forms = [("single", None)] if n == 1 else [("two_graphs", "0"), ("in_graph", "1"), ("two_buckets", "0")]
        for form, in_graph in forms:
            cenv = dict(env)
            if in_graph is not None:
                cenv["IGCN_COMM_IN_GRAPH"] = in_graph
                cenv["IGCN_DP_TWO_BUCKETS"] = "1" if form == "two_buckets" else "0"
            try:
                r = subprocess.run(cmd, env=cenv, stdout=subprocess.PIPE, text=True, timeout=tmo)
            except subprocess.TimeoutExpired:
                print(f"[bench] sweep: N={n} form={form} timed out after {tmo:.0f} s", file=sys.stderr)
                rc = rc or 5
                continue
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
            if r.returncode != 0 or not lines:
                print(f"[bench] sweep: N={n} form={form} failed (rc={r.returncode})", file=sys.stderr)
                rc = rc or r.returncode or 4
                continue
            line = json.loads(lines[-1])
            line["exchange_form"] = form
            if n == 1:
                n1 = line["value"]
            elif n1 is not None and "weak_scaling_efficiency_vs_n1" not in line:
                line["weak_scaling_efficiency_vs_n1"] = round(line["value"] / (n * n1), 4)
            print(json.dumps(line), flush=True)
    return rc


def main():
    # the image exports NCCL_DEBUG=VERSION: RCCL then prints a five-line banner on STDOUT at communicator creation —
    # in front of the one JSON line this script owes its caller
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        del os.environ["NCCL_DEBUG"]
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the roofline measurements (and the profiler child)")
    ap.add_argument("--eager", action="store_true", help="no hipGraph replay of the step")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="full",
                    help="full = BASELINE configs[2] (the metric); stress = configs[4]; sgcn = configs[1]")
    ap.add_argument("--bf16", choices=["auto", "0", "1"], default="auto",
                    help="dense feature transforms with bf16 operands (auto: the workload's own setting)")
    ap.add_argument("--rotate", type=int, default=0,
                    help="time GraphedTrainStep.load + replay over this many distinct device-resident batches")
    ap.add_argument("--pipeline", action="store_true",
                    help="loader-fed runs: default for the sparse workloads since round 4; forces them for --workload stress")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="skip the loader-fed runs (host feeder thread / device-resident dataset / device GDC)")
    ap.add_argument("--blocks", type=int, default=5,
                    help="the timed region (--steps steps) is repeated this many times; the median block is reported")
    ap.add_argument("--no-stress", action="store_true",
                    help="default workload only: skip the short configs[4] child run reported as `stress`")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0, help="bound of the CPU-oracle sample")
    ap.add_argument("--model-sweep", action="store_true",
                    help="also time the five (layers, hidden) entries of the reference's sweep (main.py:152-154)")
    ap.add_argument("--no-model-sweep", action="store_true", help="skip them in the default line")
    ap.add_argument("--inner-profile", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-cores", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_child:                           # the pinned CPU sample: no GPU call in this process
        cpu_baseline_child(args.workload, args.cpu_baseline_seconds,
                           [int(c) for c in args.cpu_baseline_cores.split(",") if c])
        return
    if os.environ.get("IGCN_BENCH_SWEEP") and "WORLD_SIZE" not in os.environ:
        sys.exit(sweep([int(v) for v in os.environ["IGCN_BENCH_SWEEP"].replace(" ", "").split(",") if v]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))                  # no launcher around us: become one (no GPU call made yet)
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    # IGCN_BENCH_ONE_DEVICE=1 (rehearsal on a single-GPU box only): every rank shares cuda:0 and the gradient
    # exchange runs over gloo, so the multi-process control flow can be exercised without several GPUs
    rehearsal = os.environ.get("IGCN_BENCH_ONE_DEVICE", "0") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # IGCN_BENCH_FORCE_DIST=1 with one rank: an RCCL process group of size 1 and the N>1 control flow on a one-GPU box
    force_dist = world == 1 and os.environ.get("IGCN_BENCH_FORCE_DIST", "0") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=device)     # backend nccl == RCCL over xGMI

    from igcn_amd import _lib
    from igcn_amd.train import FlatAdam, GraphedTrainStep, train_step
    _lib.load()                                   # fail loudly when the HIP library is missing

    wl = WORKLOADS[args.workload]
    bf16 = wl["bf16"] if args.bf16 == "auto" else args.bf16 == "1"
    per_gpu = wl["graphs"]
    model, go = build_model(device, wl, bf16)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    if world > 1:                                 # identical replicas
        torch.distributed.broadcast(opt.flat, 0)
    graphs = synth.brain_graph_list(per_gpu, seed=1000 + rank, rois=wl["rois"], tsne_dim=90, dense=wl["dense"])
    data = Batch.from_data_list(graphs).to(device)
    data.x.requires_grad_(True)

    # gradient exchange: libigcn's own RCCL communicator (igcn_comm_*: all-reduce on the launch stream, captured into
    # the step graph), verified against torch.distributed once; any failure falls back to torch.distributed's group
    comm, exchange = None, "none"
    dist_on = world > 1 or force_dist
    if dist_on and not rehearsal and os.environ.get("IGCN_BENCH_TORCH_ALLREDUCE", "0") != "1":
        try:
            from igcn_amd.comm import Comm
            comm = Comm()
            probe = torch.arange(1024, dtype=torch.float32, device=device) * (rank + 1)
            want = probe.clone()
            torch.distributed.all_reduce(want)
            comm.all_reduce_(probe)
            torch.cuda.synchronize()
            if not torch.equal(probe, want):
                raise RuntimeError("igcn_comm all-reduce disagrees with torch.distributed")
            exchange = "igcn_comm (RCCL via the C ABI)"
        except Exception as exc:                  # noqa: BLE001
            print(f"[bench] igcn_comm unavailable ({type(exc).__name__}: {exc}); using torch.distributed",
                  file=sys.stderr)
            comm = None
        if world > 1:
            # every rank takes the SAME exchange: one rank on igcn_comm and another on torch.distributed would deadlock
            flag = torch.tensor([1 if comm is not None else 0], device=device, dtype=torch.int32)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            if int(flag.item()) == 0 and comm is not None:
                comm.close()
                comm = None
    if dist_on and comm is None:
        exchange = "torch.distributed all_reduce (%s)" % torch.distributed.get_backend()

    # IGCN_DP_TWO_BUCKETS=1: the heads' gradients (4/5 of the bucket) all-reduced on a side stream while the rest of the
    # backward runs, the remainder behind it (train._TwoBucketExchange; three graphs, collectives outside them)
    two_buckets = dist_on and os.environ.get("IGCN_DP_TWO_BUCKETS", "0") == "1"

    def eager_step():
        data._igcn_plan = None                    # the plan is per batch: rebuild it inside every step
        return train_step(model, opt, data, world_size=world, comm=comm, two_buckets=two_buckets)

    launch = "eager"
    step = eager_step
    gstep = None
    if not args.eager:
        try:
            # N > 1: the all-reduce sits BETWEEN two graph replays on the launch stream by default; capturing the RCCL
            # call into the one step graph is verified on a single-rank communicator only (IGCN_COMM_IN_GRAPH=1)
            in_graph = None if os.environ.get("IGCN_COMM_IN_GRAPH", "0") == "1" else False
            gstep = step = GraphedTrainStep(model, opt, data, world_size=world, distributed=dist_on, comm=comm,
                                            comm_in_graph=in_graph, two_buckets=two_buckets)
            launch = "hipGraph replay"
            if dist_on and gstep.two is not None:
                launch += (" (three graphs: the heads' all-reduce on a side stream beside the rest of the backward, the "
                           "remainder's behind it)")
            elif dist_on:
                launch += " (all-reduce inside the graph)" if gstep.comm_in_graph else " (two graphs around the all-reduce)"
        except Exception as exc:                  # noqa: BLE001 — a capture refused by the runtime must not sink the run
            print(f"[bench] graph capture failed ({type(exc).__name__}: {exc}); running the eager step",
                  file=sys.stderr)
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.inner_profile:
        # the child under rocprofv3: exactly --steps replays between two marker launches (a kernel no step contains), so
        # the parent reads per-kernel launches and durations of REPLAYS only off the kernel trace
        from igcn_amd._lib import call, stream_ptr
        mark = torch.empty(64, 16, device=device)
        call("igcn_launch_floor", 64, 16, 0, 0, mark.data_ptr(), stream_ptr())
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        call("igcn_launch_floor", 64, 16, 0, 0, mark.data_ptr(), stream_ptr())
        torch.cuda.synchronize()
        print(json.dumps({"inner_steps": args.steps, "ms_per_step": round((time.perf_counter() - t0) / args.steps * 1e3, 3)}))
        return

    def timed_block():
        """EXACTLY --steps steps between barrier + synchronize on both sides; the max over ranks."""
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    # the timed region is repeated: --blocks blocks of --steps steps each; the line reports the MEDIAN block (and the
    # spread), so one disturbed block does not decide the headline
    block_s = []
    for _ in range(max(1, args.blocks)):
        dt, loss = timed_block()
        block_s.append(dt)
    dt = sorted(block_s)[len(block_s) // 2]
    if not bool(torch.isfinite(loss)):
        print("non-finite loss", file=sys.stderr)
        sys.exit(3)

    # the distributed step taken apart (two-graph form; an all-reduce captured into the step graph cannot be bracketed):
    # HIP events on the launch stream around [forward..backward + pack] graph | all-reduce | [Adam] graph of extra steps,
    # the host's enqueue time per step, and the all-reduce alone back to back (no step around it) — per RANK
    allreduce_us, dist_parts = None, None
    if dist_on and gstep is not None and gstep.g_opt is not None:
        def ev():
            return torch.cuda.Event(enable_timing=True)
        recs, host_us = [], []
        two = gstep.two
        for _ in range(20):
            gstep._own_the_table()
            e = [ev() for _ in range(5)]
            h0 = time.perf_counter()
            e[0].record()
            gstep.g_main.replay()
            e[1].record()
            if two is not None:                  # the heads' all-reduce runs on the side stream beside g_rest
                two.start_early()
                gstep.g_rest.replay()
                e[4].record()
                two.finish()
            else:
                e[4].record()
                gstep._reduce()
            e[2].record()
            gstep.g_opt.replay()
            e[3].record()
            host_us.append((time.perf_counter() - h0) * 1e6)
            recs.append(e)
        torch.cuda.synchronize()
        alone = []
        for _ in range(20):
            a, b = ev(), ev()
            a.record()
            if two is not None:                  # both buckets, one after the other on the launch stream
                two._all_reduce(two.early)
                for t_ in two.rest:
                    two._all_reduce(t_)
            else:
                gstep._reduce()
            b.record()
            alone.append((a, b))
        torch.cuda.synchronize()
        med = lambda v: sorted(v)[len(v) // 2]                    # noqa: E731
        mine = [med([e[0].elapsed_time(e[1]) * 1e3 for e in recs]), med([e[4].elapsed_time(e[2]) * 1e3 for e in recs]),
                med([e[2].elapsed_time(e[3]) * 1e3 for e in recs]), med([a.elapsed_time(b) * 1e3 for a, b in alone]),
                med(host_us), med([e[1].elapsed_time(e[4]) * 1e3 for e in recs])]
        t = torch.tensor(mine, device=device, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)] if world > 1 else [t]
        if world > 1:
            torch.distributed.all_gather(every, t)
        rows = [[round(float(v), 1) for v in r.tolist()] for r in every]
        totals = [r[0] + r[1] + r[2] + r[5] for r in rows]
        allreduce_us = max(r[1] for r in rows)
        dist_parts = {
            "exchange": "two_buckets" if two is not None else "two_graphs",
            "per_rank_median_us": {"g_main_replay": [r[0] for r in rows],
                                   "allreduce_in_step (incl. any wait for the host to enqueue it)": [r[1] for r in rows],
                                   "allreduce_alone_back_to_back": [r[3] for r in rows],
                                   "gap_before_allreduce = in_step - alone": [round(r[1] - r[3], 1) for r in rows],
                                   "g_opt_replay": [r[2] for r in rows],
                                   "g_rest_replay (two-bucket form: the rest of the backward, beside the heads' all-reduce)":
                                       [r[5] for r in rows],
                                   "host_enqueue_per_step": [r[4] for r in rows]},
            "slowest_rank": int(max(range(len(rows)), key=lambda i: totals[i])),
            "device_us_per_step_by_rank": [round(v, 1) for v in totals], "steps_sampled": 20}

    if rank == 0:
        total_graphs = per_gpu * world * args.steps
        res = {
            "metric": "graphs/s train step (90-ROI brain + GO-SNP)", "value": round(total_graphs / dt, 1),
            "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "timing": {"blocks": len(block_s), "steps_per_block": args.steps, "reported": "median block",
                       "ms_per_step_min": round(min(block_s) / args.steps * 1e3, 3),
                       "ms_per_step_max": round(max(block_s) / args.steps * 1e3, 3),
                       "ms_per_step_blocks": [round(b / args.steps * 1e3, 3) for b in block_s]},
            "vs_baseline": None, "dtype": "bf16" if (bf16 and wl["pool"] is not None) else "f32", "data": "synthetic",
            "config": {"workload": wl["name"],
                       "graphs_per_gpu": per_gpu, "global_batch": per_gpu * world,
                       "layers": LAYERS, "hidden": HIDDEN, "rois": wl["rois"],
                       "go_nodes": sum(wl["pool"]) if wl["pool"] else 0,
                       "parallelism": f"dp{world}", "launch": launch, "gradient_exchange": exchange,
                       "rccl_world_size": torch.distributed.get_world_size() if dist_on else 1},
            "loss": round(float(loss), 6),
        }
        if dist_on:
            res["config"]["allreduce_us_per_step"] = allreduce_us      # median over 20 steps, max over ranks
            res["config"]["allreduce_bytes"] = int(opt.grad.numel()) * 4
            if dist_parts is not None:
                res["distributed_step"] = dist_parts
            n1 = os.environ.get("IGCN_BENCH_N1_VALUE")                  # graphs/s of the N = 1 run, when the caller has it
            if n1:
                res["weak_scaling_efficiency_vs_n1"] = round(res["value"] / (world * float(n1)), 4)
        if args.rotate > 1 and gstep is not None:
            # hand-over of NEW batches: GraphedTrainStep.load (device-to-device copies into the static inputs) + replay
            pool_b = [Batch.from_data_list(synth.brain_graph_list(per_gpu, seed=2000 + i, rois=wl["rois"], tsne_dim=90,
                                                                  dense=wl["dense"])).to(device)
                      for i in range(args.rotate)]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                gstep.load(pool_b[i % args.rotate])
                gstep()
            torch.cuda.synchronize()
            res["rotating_batches"] = {"batches": args.rotate,
                                       "ms_per_step_load_plus_replay": round((time.perf_counter() - t1) / args.steps * 1e3, 3)}
        # (dense 512-ROI subjects are 1.3 MB of edge list each: that dataset is not host-fed per step; --pipeline forces it)
        if not args.no_pipeline and gstep is not None and world == 1 and wl["pool"] is not None \
                and (not wl["dense"] or args.pipeline):
            # the loader-fed step (the reference's step starts at `for data in loader: data = data.to(device)`, :515-517)
            res["pipeline"] = pipeline_bench(gstep, wl, device, max(args.steps, 60), args.warmup, res["ms_per_step"])
        if wl["pool"] is not None and world == 1 and not args.no_roofline:
            stats = instep_profile(args.workload, bf16)
            dense_rf = dense_roofline(data, wl, stats) if wl["dense"] else None
            standalone = scatter_roofline(data, device, wl, stats) if dense_rf is None else None
            fused = None
            if dense_rf is None and (stats is None or _pick(stats[0], "k_gcn_propagate_fwd") is None):  # (front / stack kernel)
                fused = fused_stack_roofline(model, data, device, wl, stats)
            if dense_rf is not None:
                res["roofline"] = dense_rf
            elif fused is not None:
                # the default step runs the scatter-aggregate inside the LDS-resident stack kernel; the stand-alone
                # kernel (other widths / graph shapes) keeps its own replay measurements beside it
                res["roofline"] = fused
                standalone["timing"] = "not launched by this step (fused into k_sgcn_stack_fwd); " + standalone["timing"]
                res["roofline_scatter_standalone"] = standalone
            else:
                res["roofline"] = standalone
            res["roofline_mfma"] = mfma_roofline(model, wl, device, stats, bf16)
            if stats:
                lib = [(c, t) for n, (c, _, t) in stats[0].items() if (n[5:] if n.startswith("void ") else n).startswith("k_")]
                res["profile"] = {"replays_profiled": stats[1], "scope": "the child's timed hipGraph replays only "
                                  "(bracketed by marker launches in the kernel trace)",
                                  "kernel_us_per_step": round(sum(t for _, _, t in stats[0].values()) / stats[1], 1),
                                  "launches_per_step": round(sum(c for c, _, _ in stats[0].values()) / stats[1], 1),
                                  "libigcn_launches_per_step": round(sum(c for c, _ in lib) / stats[1], 1),
                                  "span_us_per_step": round(stats[3] / stats[1], 1)}
                whole = step_traffic(args.workload, res["profile"]["kernel_us_per_step"])
                if whole is not None:
                    res["roofline_step"] = whole
        if wl["pool"] is not None and world == 1 and not wl["dense"] and gstep is not None \
                and (args.model_sweep or (args.workload == "full" and not args.no_roofline and not args.no_model_sweep)):
            res["model_sweep"] = model_sweep(wl, device, data)
        if wl["pool"] is not None and world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.workload, seconds=args.cpu_baseline_seconds, device=device)
        if args.workload == "full" and world == 1 and not args.no_roofline and not args.no_stress:
            # the device is idle from here on: this process has finished its own measurements
            torch.cuda.synchronize()
            st = stress_child()
            if st is not None:
                res["stress"] = st
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
    if comm is not None:
        comm.close()
    if world > 1 or force_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
