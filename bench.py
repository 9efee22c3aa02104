#!/usr/bin/env python3
"""bench.py — graphs/s of the full IG-GCN train step (SGCN over 90-ROI brain graphs + GO-SNP network).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2]): full SGCN_GCN_IMGSNP (L=2, hidden=16, R=90, H0=3,
cross-attention fusion, 3 classes, 3 regression targets), synthetic GO DAG N=3000 pool [1800,800,300,99,1],
256 graphs per GPU (weak scaling), fp32.  One step = kernel/train_eval_sgcn_img_snps.py:515-547: graph-plan
build, forward, masked forward, 7 loss terms, backward, (all-reduce), Adam — on inputs already resident
in HBM.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
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
# HBM bytes per launch from the rocprofv3 PMC passes of tools/roofline_kernel.py (FETCH_SIZE x2 gfx950 correction,
# calibrated on a 256 MiB float4 copy; WRITE_SIZE x1): profiles/r01_pmc/scatter_aggregate_traffic.json
PMC_TRAFFIC = {"bench": 10010260, "stress": 77631040}
# average duration of the same kernels in the committed rocprofv3 --kernel-trace --stats summary of this command
# (profiles/r01_final_default_bench/kernel_stats.csv).  The profiler adds ~1-2 us to every dispatch and sees the
# in-step launches with cold caches, which matters for the 3-us kernel and not for the 36-us one.
ROCPROF_AVG_US = {"bench": 4.47, "stress": 35.8}


# --workload: the default is the configuration the metric is quoted on; the other two are side measurements
WORKLOADS = {
    "full": dict(rois=ROIS, pool=POOL, graphs=GRAPHS_PER_GPU, dense=False,
                 name="configs[2]: full sgcn_img_snp train step (2 fwd + 7 losses + bwd + Adam), "
                      "90-ROI k=3 brain graphs + 3000-node GO-SNP DAG"),
    "sgcn": dict(rois=ROIS, pool=None, graphs=GRAPHS_PER_GPU, dense=False,
                 name="configs[1]: SGCN-only train step (kernel/train_eval_sgcn.py:296-314), 90-ROI k=3 brain graphs"),
    "stress": dict(rois=512, pool=(6000, 2700, 1000, 299, 1), graphs=32, dense=True,
                   name="configs[4] shape in fp32: full train step, 512-ROI dense brain graphs + 10k-node GO DAG"),
}


def build_model(device, wl=None):
    wl = wl or WORKLOADS["full"]
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
                            isuseProb4Regr=True, isImageOnly=False, isSNPsOnly=False).to(device)
    model.train()
    return model, (go_snps, adj, pool_dim)


def _time_propagate(plan, ew, n, f, device, iters, nodes_per_graph=0):
    """Average device time (us) of igcn_gcn_propagate_fwd: `iters` back-to-back launches captured in one hipGraph
    and bracketed by two HIP events on the launch stream."""
    from igcn_amd import ops
    from igcn_amd._lib import call, stream_ptr
    coef = ops.GcnNorm.apply(ew, plan)
    h = torch.randn(n, f, device=device)
    bias = torch.zeros(f, device=device)
    out = torch.empty_like(h)
    args = (n, plan.n_edges, f, nodes_per_graph, h.data_ptr(), f, coef[2].data_ptr(), coef[1].data_ptr(),
            bias.data_ptr(), plan.tgt_ptr.data_ptr(), out.data_ptr(), f, 1)
    for _ in range(5):
        call("igcn_gcn_propagate_fwd", *args, stream_ptr())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):      # a process group's threads may be alive
        for _ in range(iters):
            call("igcn_gcn_propagate_fwd", *args, stream_ptr())
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def scatter_roofline(data, device, iters=200):
    """Live roofline of the GCN scatter-aggregate kernel (igcn_gcn_propagate_fwd, F=16) exactly as the train step
    launches it: both passes batched = 2 copies of the batch's graphs.  Algorithmic bytes = (20*E' + 8*R*F) per
    graph (int64 endpoints + fp32 coefficient per edge, each feature row read and written once)."""
    from igcn_amd import ops
    plan = ops.plan_for(data).replicate(2)
    n, f = 2 * data.x.shape[0], HIDDEN
    us = _time_propagate(plan, torch.cat([data.edge_attr, data.edge_attr]), n, f, device, iters)
    n_graphs = n // ROIS
    e_prime = plan.n_edges // n_graphs          # GDC graphs store their self-loops: E' = E
    alg_bytes = n_graphs * (20 * e_prime + 8 * ROIS * f)
    gbs = alg_bytes / (us * 1e-6) / 1e9
    return {"bound": "hbm", "kernel": "k_gcn_propagate_fwd_q<4>", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": PMC_TRAFFIC["bench"] if (n_graphs, e_prime, f) == (512, 270, 16) else None,
            "alg_bytes_per_launch": alg_bytes, "us_per_launch": round(us, 3),
            "rocprof_avg_us": ROCPROF_AVG_US["bench"] if (n_graphs, e_prime, f) == (512, 270, 16) else None,
            "launch": f"{n_graphs} graphs x {e_prime} edges (both passes of a step), F={f}"}


def scatter_roofline_stress(device, n_graphs=32, rois=512, f=16, iters=20):
    """The same kernel at the stress shape of BASELINE.json configs[4] (512-ROI dense graphs, E' = R^2 per graph):
    the launch moves ~170 MB, so it is bandwidth- rather than latency-limited."""
    import numpy as np
    from igcn_amd import ops
    rng = np.random.default_rng(0)
    r = torch.arange(rois).repeat_interleave(rois)
    c = torch.arange(rois).repeat(rois)
    ei = torch.cat([torch.stack([r, c]) + g * rois for g in range(n_graphs)], dim=1).to(device)
    w = torch.from_numpy(rng.random(ei.shape[1]).astype(np.float32) / rois).to(device)
    n = n_graphs * rois
    plan = ops.GraphPlan(ei, n)
    us = _time_propagate(plan, w, n, f, device, iters, nodes_per_graph=rois)
    e_prime = rois * rois
    alg_bytes = n_graphs * (20 * e_prime + 8 * rois * f)
    gbs = alg_bytes / (us * 1e-6) / 1e9
    return {"bound": "hbm", "kernel": "k_gcn_propagate_fwd_wide<4>", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": PMC_TRAFFIC["stress"] if (n_graphs, rois, f) == (32, 512, 16) else None,
            "alg_bytes_per_launch": alg_bytes, "us_per_launch": round(us, 3),
            "rocprof_avg_us": ROCPROF_AVG_US["stress"] if (n_graphs, rois, f) == (32, 512, 16) else None,
            "launch": f"{n_graphs} dense graphs x {rois} ROIs ({e_prime} edges each), F={f}"}


def cpu_baseline(go, seconds=20.0):
    """The oracle (CPU restatement, faithful mode: per-sample sparse loop) timed on this box's host cores on
    a bounded sample of the same workload: B=32 graphs per step."""
    from types import SimpleNamespace
    from oracle import go_network as OG, sgcn_img_snp as OS
    go_snps, adj, pool_dim = go
    a_g, a = synth.go_sparse_inputs(go_snps, adj)
    idx = OG.go_index_sets(a_g, a, list(POOL), 2)
    shapes = dict(OS.sgcn_param_shapes(LAYERS, HIDDEN, rois=ROIS))
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
    cfg = SimpleNamespace(num_layers=LAYERS, rois=ROIS, image_only=False, rbf_gamma=0.01)
    b = 32
    data = Batch.from_data_list(synth.brain_graph_list(b, seed=1000, rois=ROIS, tsne_dim=90))
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    opt = None
    times = []
    t_end = time.perf_counter() + seconds
    it = 0
    while it < 2 or (time.perf_counter() < t_end and it < 12):
        t0 = time.perf_counter()
        _, _, opt = OS.train_step(sd, cfg, idx, data, lr=1e-3, dropout=True, faithful=True, opt=opt)
        times.append(time.perf_counter() - t0)
        it += 1
    times = sorted(times[1:]) if len(times) > 1 else times
    med = times[len(times) // 2]
    return {"value": round(b / med, 2), "unit": "graphs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} train steps of B={b} graphs (same model/GO DAG), median; oracle faithful mode"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="no hipGraph replay of the step")
    ap.add_argument("--no-stress", action="store_true", help="skip the stress-shape roofline measurement")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="full",
                    help="full = BASELINE configs[2] (the metric); sgcn / stress are side measurements")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("launch multi-GPU runs with torch.distributed.run (one process per GPU)", file=sys.stderr)
        sys.exit(2)
    # IGCN_BENCH_ONE_DEVICE=1 (rehearsal on a single-GPU box only): every rank shares cuda:0 and the gradient
    # exchange runs over gloo, so the multi-process control flow can be exercised without several GPUs
    rehearsal = os.environ.get("IGCN_BENCH_ONE_DEVICE", "0") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # IGCN_BENCH_FORCE_DIST=1 with one rank: an RCCL process group of size 1 and the N>1 control flow (two graphs
    # around the all-reduce) on a one-GPU box
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
    per_gpu = wl["graphs"]
    model, go = build_model(device, wl)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    if world > 1:                                 # identical replicas
        torch.distributed.broadcast(opt.flat, 0)
    graphs = synth.brain_graph_list(per_gpu, seed=1000 + rank, rois=wl["rois"], tsne_dim=90, dense=wl["dense"])
    data = Batch.from_data_list(graphs).to(device)
    data.x.requires_grad_(True)

    def eager_step():
        data._igcn_plan = None                    # the plan is per batch: rebuild it inside every step
        return train_step(model, opt, data, world_size=world)

    launch = "eager"
    step = eager_step
    if not args.eager:
        try:
            step = GraphedTrainStep(model, opt, data, world_size=world, distributed=world > 1 or force_dist)
            launch = "hipGraph replay"                  # the whole step (N>1: two graphs around the all-reduce)
        except Exception as exc:                  # noqa: BLE001 — a capture refused by the runtime must not sink the run
            print(f"[bench] graph capture failed ({type(exc).__name__}: {exc}); running the eager step",
                  file=sys.stderr)
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if not bool(torch.isfinite(loss)):
        print("non-finite loss", file=sys.stderr)
        sys.exit(3)

    if rank == 0:
        total_graphs = per_gpu * world * args.steps
        res = {
            "metric": "graphs/s train step (90-ROI brain + GO-SNP)", "value": round(total_graphs / dt, 1),
            "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"],
                       "graphs_per_gpu": per_gpu, "global_batch": per_gpu * world,
                       "layers": LAYERS, "hidden": HIDDEN, "rois": wl["rois"],
                       "go_nodes": sum(wl["pool"]) if wl["pool"] else 0,
                       "parallelism": f"dp{world}", "launch": launch},
            "loss": round(float(loss), 6),
        }
        if args.workload == "full":
            res["roofline"] = scatter_roofline(data, device)
            if world == 1 and not args.no_stress:
                res["roofline_stress"] = scatter_roofline_stress(device)
            if world == 1 and not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(go)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
    if world > 1 or force_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
