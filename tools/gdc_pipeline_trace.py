#!/usr/bin/env python3
"""Kernel trace of the GDC-fed train step (loader.DeviceGdcFeeder -> GraphedTrainStep.load -> replay).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 <repo>/tools/gdc_pipeline_trace.py [steps]
    python3 tools/replay_trace.py --summarise <dir> [<out.csv>]

K fed steps between two marker launches (k_launch_floor): the summary lists, per step, the launches of the side
stream's chain (index_select, k_gdc_topk, cumsum, the copies into the slot), of ``load`` and of the replay."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from igcn_amd import synth
    from igcn_amd._lib import call, stream_ptr
    from igcn_amd.data import Batch, Data
    from igcn_amd.loader import DeviceGdcFeeder, UniformGraphStore
    from igcn_amd.train import FlatAdam, GraphedTrainStep
    dev = torch.device("cuda", 0)
    wl = bench.WORKLOADS["full"]
    b = wl["graphs"]
    model, _ = bench.build_model(dev, wl)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    graphs = synth.brain_graph_list(2 * b, seed=3000, rois=wl["rois"], tsne_dim=90)
    data = Batch.from_data_list(graphs[:b]).to(dev)
    data.x.requires_grad_(True)
    step = GraphedTrainStep(model, opt, data)
    keep = ("x", "edge_index", "edge_attr", "snps_feat", "y", "clini_score", "tsne_fdim", "clust_y")
    store = UniformGraphStore([Data(**{n: getattr(g, n) for n in keep}) for g in graphs], dev)
    adj = torch.stack([g.A for g in graphs]).to(dev)
    cols = {n: store.cols[n] for n in ("x", "snps_feat", "y", "clini_score", "tsne_fdim", "clust_y")}
    cur = torch.cuda.current_stream()
    feed = iter(DeviceGdcFeeder(adj, cols, b, k + 4, seed=1))

    def consume(batch):
        cur.wait_event(batch.ready)
        step.load(batch)
        batch.release()
        step()
    for _ in range(3):
        consume(next(feed))
    torch.cuda.synchronize()
    mark = torch.empty(64, 16, device=dev)
    call("igcn_launch_floor", 64, 16, 0, 0, mark.data_ptr(), stream_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        consume(next(feed))
    e1.record()
    call("igcn_launch_floor", 64, 16, 0, 0, mark.data_ptr(), stream_ptr())
    torch.cuda.synchronize()
    out = os.environ.get("IGCN_REPLAY_META")
    if out:
        with open(out, "w") as fh:
            fh.write(f"{k} replays gdc-fed\n")
    print(f"{k} GDC-fed steps: {e0.elapsed_time(e1) / k * 1e3:.1f} us per step, loss {float(step.loss):.6f}")


if __name__ == "__main__":
    main()
