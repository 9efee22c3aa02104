"""``python bench.py --gpus N`` as the driver would type it, without a launcher around it (BASELINE configs[3] is
run that way or under torch.distributed.run).  bench.py then starts its own ranks as child processes.

* CPU: on a box without N GPUs the command says what is missing and exits 2 (no traceback, no GPU call).
* GPU: the one-device rehearsal (IGCN_BENCH_ONE_DEVICE=1: both ranks share cuda:0, gradients over gloo) runs the
  multi-rank control flow end to end — rendezvous, parameter broadcast, shard-seeded batches, graphed step around the
  all-reduce, barrier-bracketed timing, max over ranks, ONE JSON line from rank 0.
"""
import json
import math
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def test_gpus_n_without_devices_exits_2_with_a_message():
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "IGCN_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert "IGCN_BENCH_ONE_DEVICE" in r.stderr and "Traceback" not in r.stderr
    assert r.stdout.strip() == ""


@pytest.mark.gpu
def test_gpus_2_self_launch_rehearsal_on_one_device():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(IGCN_BENCH_ONE_DEVICE="1", IGCN_BENCH_N1_VALUE="250000")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-roofline",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["scaling"] == "weak"
    cfg = res["config"]
    assert cfg["rccl_world_size"] == 2 and cfg["global_batch"] == 2 * cfg["graphs_per_gpu"]
    assert cfg["parallelism"] == "dp2" and "all_reduce" in cfg["gradient_exchange"]
    assert cfg["allreduce_us_per_step"] is not None and cfg["allreduce_us_per_step"] > 0
    assert math.isfinite(res["loss"]) and res["value"] > 0
    assert res["weak_scaling_efficiency_vs_n1"] == pytest.approx(res["value"] / 500000.0, abs=1e-4)
    parts = res["distributed_step"]                              # the N > 1 step taken apart, per rank
    assert parts["slowest_rank"] in (0, 1) and len(parts["device_us_per_step_by_rank"]) == 2
    for key, vals in parts["per_rank_median_us"].items():
        assert len(vals) == 2 and all(math.isfinite(v) for v in vals), key
    assert res["timing"]["blocks"] == 5 and len(res["timing"]["ms_per_step_blocks"]) == 5
    assert res["timing"]["ms_per_step_min"] <= res["ms_per_step"] <= res["timing"]["ms_per_step_max"]


@pytest.mark.gpu
def test_sweep_prints_one_line_per_n_and_exchange_form_with_efficiency():
    """IGCN_BENCH_SWEEP="1,2": one invocation, a fresh child per N and — at N > 1 — per gradient-exchange form
    (``two_graphs`` around the all-reduce; ``two_buckets``: three graphs, the heads' all-reduce on a side stream beside the
    rest of the backward; ``in_graph``: the collective captured, which on this one-device rehearsal
    (gloo, no RCCL communicator) is refused and falls back on every rank alike — the control flow a multi-rank run
    takes when the capture is refused), every N = 2 line scaled by the sweep's own N = 1 run."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "IGCN_BENCH_N1_VALUE")}
    env.update(IGCN_BENCH_ONE_DEVICE="1", IGCN_BENCH_SWEEP="1,2")
    r = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "1", "--blocks", "2", "--no-roofline",
                        "--no-cpu-baseline", "--no-pipeline", "--no-stress"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.strip()]
    assert [ln["n_gpus"] for ln in lines] == [1, 2, 2, 2]
    assert [ln["exchange_form"] for ln in lines] == ["single", "two_graphs", "in_graph", "two_buckets"]
    assert "three graphs" in lines[3]["config"]["launch"]
    for ln in lines[1:]:                            # the N > 1 step taken apart, with the exchange it ran named
        assert ln["distributed_step"]["exchange"] == ("two_buckets" if ln["exchange_form"] == "two_buckets" else "two_graphs")
    assert "weak_scaling_efficiency_vs_n1" not in lines[0]
    for ln in lines[1:]:
        assert ln["weak_scaling_efficiency_vs_n1"] == pytest.approx(ln["value"] / (2 * lines[0]["value"]), abs=1e-3)
        assert "launch" in ln["config"]
