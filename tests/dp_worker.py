"""Child process of tests/test_dp_hip.py: ONE rank of a world_size-W data-parallel train step on the HIP path.
Every rank shares cuda:0 (a one-GPU box) and the gradient exchange runs over gloo — RCCL refuses two ranks on one
device — so what is exercised is the product's own multi-rank code: shard_batch, igcn_pack_grads (table -> flat
bucket), the all-reduce, igcn_adam_step(from_flat, grad_scale = 1/W), eagerly and as GraphedTrainStep(distributed).

    python tests/dp_worker.py RANK WORLD PORT OUT_PREFIX
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import igcn_amd  # noqa: E402,F401
from igcn_amd import synth  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402

# real brain-graph dims (R = 90, hidden 16: the LDS-resident fused SGCN stack at F = 16, the MFMA attention core) and a
# 320-node GO DAG (LDS-resident GO attention backward and decoder) — the kernels of the default step, on two ranks
POOL = (200, 80, 30, 9, 1)
ROIS, HIDDEN, LAYERS, N_GRAPHS, SEED = 90, 16, 2, 32, 3
LAM = [1.0, 1.0, 0.5, 1.5e-6, 0.1, 0.2]


def build_model(device):
    from _weights import seeded_state
    from igcn_amd.sgcn_img_snp import SGCN_GCN_IMGSNP
    go_snps, adj, pool_dim = synth.go_hierarchy(POOL, seed=2)
    a_g, a = synth.go_sparse_inputs(go_snps, adj, device)
    model = SGCN_GCN_IMGSNP(LAYERS, HIDDEN, a_g, a, pool_dim, 32, device, rois=ROIS, H_0=3, num_classes=3,
                            isSoftSimilarity=True, rbf_gamma=0.01, isCrossAtten=True, num_regr=3,
                            isuseProb4Regr=True, isImageOnly=False, isSNPsOnly=False).to(device)
    sd = seeded_state({k: v.shape for k, v in model.state_dict().items()}, SEED, model.state_dict())
    model.load_state_dict(sd)
    model.train()
    model._dropout_enabled = False
    model.go_network._dropout_enabled = False
    return model, (go_snps, adj)


def all_graphs():
    return synth.brain_graph_list(N_GRAPHS, seed=4, rois=ROIS, tsne_dim=6)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from igcn_amd import _lib
        from igcn_amd.train import FlatAdam, GraphedTrainStep, shard_batch, train_step
        _lib.load()
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        graphs = shard_batch(all_graphs(), rank, world)
        res = {}
        # (A) eager step
        model, _ = build_model(dev)
        names = [k for k, p in model.named_parameters() if p.requires_grad]
        opt = FlatAdam(model.parameters(), lr=1e-3)
        data = Batch.from_data_list(graphs).to(dev)
        loss = train_step(model, opt, data, LAM, world_size=world)
        torch.cuda.synchronize()
        res["loss"] = float(loss)
        res["grad_sum"] = {k: opt.grad[o:o + p.numel()].view_as(p).cpu().clone()
                           for k, o, p in zip(names, opt._offs, opt.params)}
        res["param_after"] = {k: p.detach().cpu().clone() for k, p in zip(names, opt.params)}
        res["buffers"] = {k: v.cpu().clone() for k, v in model.state_dict().items() if "running" in k}
        # (B) the same step as hipGraph replays around the all-reduce
        model2, _ = build_model(dev)
        opt2 = FlatAdam(model2.parameters(), lr=1e-3)
        data2 = Batch.from_data_list(graphs).to(dev)
        data2.x.requires_grad_(True)
        step = GraphedTrainStep(model2, opt2, data2, LAM, world_size=world, distributed=True)
        loss2 = step()
        torch.cuda.synchronize()
        res["graphed_loss"] = float(loss2)
        res["graphed_param_after"] = {k: p.detach().cpu().clone() for k, p in zip(names, opt2.params)}
        res["graphed_step_count"] = int(opt2.step_count.item())
        # (C) the reference's epoch loop on every rank's shard (fit_epoch: captured steps for the full batches, the eager
        # step for the ragged tail, the rate halved between the epochs) against the same steps through train_step
        import copy
        from igcn_amd.data import DataLoader
        from igcn_amd.train import fit_epoch
        model3, _ = build_model(dev)
        model4 = copy.deepcopy(model3)
        opt3, opt4 = FlatAdam(model3.parameters(), lr=1e-3), FlatAdam(model4.parameters(), lr=1e-3)
        loader = DataLoader(graphs, 6, shuffle=False)                      # 16 graphs per rank: 6 + 6 + 4
        ep = []
        for epoch in range(3):
            ep.append(fit_epoch(model3, opt3, loader, None, LAM, device=dev, world_size=world))
            for data in loader:
                train_step(model4, opt4, data.to(dev), LAM, world_size=world)
            if epoch == 0:
                for o in (opt3, opt4):
                    for group in o.param_groups:
                        group['lr'] = 0.5 * group['lr']
        torch.cuda.synchronize()
        tr = next(iter(opt3._igcn_epoch_trainers.values()))
        res["epoch_counts"] = dict(tr.counts)
        res["epoch_losses"] = [float(v) for v in ep]
        res["epoch_param_after"] = {k: p.detach().cpu().clone() for k, p in zip(names, opt3.params)}
        res["epoch_ref_param_after"] = {k: p.detach().cpu().clone() for k, p in zip(names, opt4.params)}
        res["epoch_step_count"] = (int(opt3.step_count.item()), int(opt4.step_count.item()))
        # (D) the two-bucket exchange (heads' gradients all-reduced on a side stream beside the rest of the backward), eager
        # and as three graphs: the same gradients and parameters as (A) / (B), bit for bit
        model5, _ = build_model(dev)
        opt5 = FlatAdam(model5.parameters(), lr=1e-3)
        data5 = Batch.from_data_list(graphs).to(dev)
        loss5 = train_step(model5, opt5, data5, LAM, world_size=world, two_buckets=True)
        torch.cuda.synchronize()
        res["two_loss"] = float(loss5)
        res["two_used"] = getattr(opt5, "_two_bucket_exchange", None) is not None
        res["two_grad_sum"] = {k: opt5.grad[o:o + p.numel()].view_as(p).cpu().clone()
                               for k, o, p in zip(names, opt5._offs, opt5.params)}
        res["two_param_after"] = {k: p.detach().cpu().clone() for k, p in zip(names, opt5.params)}
        model6, _ = build_model(dev)
        opt6 = FlatAdam(model6.parameters(), lr=1e-3)
        data6 = Batch.from_data_list(graphs).to(dev)
        data6.x.requires_grad_(True)
        step6 = GraphedTrainStep(model6, opt6, data6, LAM, world_size=world, distributed=True, two_buckets=True)
        res["two_graphed_loss"] = float(step6())          # (read before the next replay overwrites the tensor)
        res["two_graphed_param_after"] = {k: p.detach().cpu().clone() for k, p in zip(names, opt6.params)}
        step6()
        torch.cuda.synchronize()
        res["two_graphed_used"] = step6.two is not None and step6.g_rest is not None
        res["two_graphed_param_after_2_steps"] = {k: p.detach().cpu().clone() for k, p in zip(names, opt6.params)}
        step()                                            # (B)'s second step, for the comparison after two steps
        torch.cuda.synchronize()
        res["graphed_param_after_2_steps"] = {k: p.detach().cpu().clone() for k, p in zip(names, opt2.params)}
        res["two_early_share"] = float(step6.two.early.numel()) / float(opt6.grad.numel())
        torch.save(res, f"{out}.rank{rank}.pt")
        torch.distributed.barrier()
    finally:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
