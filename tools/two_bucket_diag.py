import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29571")
torch.distributed.init_process_group("gloo", rank=0, world_size=1)
import dp_worker as W
from igcn_amd.data import Batch
from igcn_amd.train import FlatAdam, GraphedTrainStep
dev = torch.device("cuda", 0)
graphs = W.all_graphs()[:16]
out = {}
VARIANTS = {"two_norider": "IGCN_NO_GRAM_RIDER", "two_nogroups": "IGCN_NO_GEMM_GROUPS", "two_noheadloss": "IGCN_NO_HEAD_LOSS_FUSED", "two_nodefer": "IGCN_NO_DEFER"}
for tag, two in [("default", False), ("two", True)] + [(k, True) for k in VARIANTS]:
    for v in VARIANTS.values():
        os.environ.pop(v, None)
    if tag in VARIANTS:
        os.environ[VARIANTS[tag]] = "1"
    model, _ = W.build_model(dev)
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    opt = FlatAdam(model.parameters(), lr=1e-3)
    data = Batch.from_data_list(graphs).to(dev)
    data.x.requires_grad_(True)
    step = GraphedTrainStep(model, opt, data, W.LAM, world_size=1, distributed=True, two_buckets=two)
    step()
    torch.cuda.synchronize()
    out[tag] = {k: (g.detach().cpu().clone() if g is not None else None) for k, g in zip(names, step._grads)}
    print(tag, "two" if step.two is not None else "plain")
for a, b in [("default", "two")] + [("default", k) for k in VARIANTS]:
    bad = {}
    for k in out[a]:
        ga, gb = out[a][k], out[b][k]
        if ga is None or gb is None:
            if not (ga is None and gb is None): bad[k] = "None mismatch"
            continue
        if not torch.equal(ga, gb):
            bad[k] = (float((ga - gb).abs().max()), float(ga.abs().max()))
    print(a, "vs", b, {k: v for k, v in bad.items() if not isinstance(v, tuple) or v[0] > 1e-4 * v[1]})
    k = "go_network.t_D.0"
    ga, gb = out[a][k].flatten(), out[b][k].flatten()
    d = (ga != gb)
    print("  elements", ga.numel(), "differing", int(d.sum()), "first idx", d.nonzero()[:8].flatten().tolist())
    print("  default", ga[d][:6].tolist(), "\n  other  ", gb[d][:6].tolist())
torch.distributed.destroy_process_group()
