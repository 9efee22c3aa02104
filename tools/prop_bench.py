#!/usr/bin/env python3
"""Repeat bench.py's live roofline measurement of the scatter-aggregate kernel (bench shape) and print every sample."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from igcn_amd import synth  # noqa: E402
from igcn_amd.data import Batch  # noqa: E402

dev = torch.device("cuda", 0)
data = Batch.from_data_list(synth.brain_graph_list(256, seed=1000, rois=90, tsne_dim=90)).to(dev)
us = sorted(bench.scatter_roofline(data, dev, iters=int(os.environ.get("ITERS", "200")))["us_per_launch"]
            for _ in range(int(os.environ.get("REPS", "9"))))
print("us per launch:", us, "median", us[len(us) // 2])
