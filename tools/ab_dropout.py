import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from igcn_amd import synth
from igcn_amd.data import Batch
from igcn_amd.train import FlatAdam, GraphedTrainStep
dev = torch.device("cuda", 0)
for drop in (True, False):
    model, go = bench.build_model(dev)
    model._dropout_enabled = drop
    model.go_network._dropout_enabled = drop
    opt = FlatAdam(model.parameters(), lr=1e-3)
    data = Batch.from_data_list(synth.brain_graph_list(256, seed=1000, rois=90, tsne_dim=90)).to(dev)
    data.x.requires_grad_(True)
    step = GraphedTrainStep(model, opt, data)
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize()
    print("dropout", drop, (time.perf_counter() - t0) / 200 * 1e3, "ms")
