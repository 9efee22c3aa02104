"""Deterministic weights/inputs for the golden fixtures (numpy Generator => platform independent).

Fixtures never store weights: both the capture script (which loads them into the *reference* modules
with load_state_dict) and the tests (which load them into the oracle / the HIP model) regenerate them
from (shapes, seed) with this function.
"""
import numpy as np
import torch


def seeded_state(shapes, seed, reference_state=None):
    """name -> tensor.  ``shapes``: dict name -> shape (insertion order irrelevant: keys are sorted).

    Norm-layer scale/shift get non-trivial values too (so their gradients are exercised); running
    statistics start at the framework defaults; integer buffers are zero.
    """
    rng = np.random.default_rng(seed)
    sd = {}
    for k in sorted(shapes):
        s = tuple(shapes[k])
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_mean"):
            sd[k] = torch.zeros(s)
        elif k.endswith("running_var"):
            sd[k] = torch.ones(s)
        elif "t" in k.split(".") or "t_D" in k.split("."):
            sd[k] = torch.from_numpy(1.0 + 0.1 * rng.standard_normal(s)).float()
        elif _is_norm_scale(k, s):
            sd[k] = torch.from_numpy(1.0 + 0.2 * rng.standard_normal(s)).float()
        elif k.endswith(".bias") or k.endswith("_bias") and len(s) == 1:
            sd[k] = torch.from_numpy(0.1 * rng.standard_normal(s)).float()
        else:
            fan_in = s[-1] if len(s) > 1 else max(s[0], 1)
            sd[k] = torch.from_numpy(rng.uniform(-1, 1, s) / np.sqrt(fan_in)).float()
    if reference_state is not None:
        assert set(reference_state) == set(sd), (sorted(set(reference_state) ^ set(sd)))
        for k, v in reference_state.items():
            assert tuple(v.shape) == tuple(sd[k].shape), (k, v.shape, sd[k].shape)
    return sd


def _is_norm_scale(k, s):
    if len(s) != 1 or not k.endswith(".weight"):
        return False
    tail = k.split(".")
    return any(t in ("G_B", "G_B_D", "B", "B_D") for t in tail) or \
        any(tag in k for tag in ("conc_for_attention.1.", "latent.1.", "latent.5.", "classification.0.",
                                 "batch_norm"))


def summarise(t, seed=7):
    """Compact signature of a big tensor: (sum, abs-sum, dot with a seeded +-1 vector)."""
    t = t.detach().double().reshape(-1)
    rng = np.random.default_rng(seed)
    sign = torch.from_numpy(rng.integers(0, 2, t.numel()) * 2.0 - 1.0)
    return np.array([float(t.sum()), float(t.abs().sum()), float((t * sign).sum())])
