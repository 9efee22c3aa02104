"""oracle/dropout.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of libigcn's dropout-mask generator (``ig-gcn_amd/csrc/dropout.h``: ``dropout_masks_body``), the
replacement of the reference's ``F.dropout`` / ``nn.Dropout`` / ``nn.Dropout2d`` draws (kernel/go_model.py:124-128,246-251,
270-275; kernel/sgcn_img_snp.py:289-290,300-301).  The reference draws from torch's global generator, so there is nothing
of ITS numbers to pin; what this pins is the generator's contract — a factor is a pure function of (stream counter,
flat element index, the site's p): ``keep(i) = 0 if u(c, i) < p else 1 / (1 - p)`` with

    k0 = H(lo32(c) ^ 0x9E3779B9),  k1 = H(hi32(c) + 0x85EBCA6B + k0)
    h  = H((lo32(i) * 0x9E3779B1) ^ k0) + hi32(i) * 0x85EBCA77
    u  = (H(h ^ k1) >> 8) / 2^24                               H = lowbias32

so that a consumer may compute it instead of loading it and a test can rebuild any step's masks on the
host.  Sites are laid out back to back, each starting on a multiple of four elements.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def lowbias32(x):
    x = np.asarray(x, dtype=np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def uniforms(counter, n):
    """u(c, i) for i in [0, n): float32 in [0, 1), exactly the kernel's."""
    c = np.uint64(counter)
    k0 = lowbias32((c & M32) ^ np.uint64(0x9E3779B9))
    k1 = lowbias32(((c >> np.uint64(32)) + np.uint64(0x85EBCA6B) + k0) & M32)
    i = np.arange(n, dtype=np.uint64)
    h = (lowbias32((((i & M32) * np.uint64(0x9E3779B1)) & M32) ^ k0) + (i >> np.uint64(32)) * np.uint64(0x85EBCA77)) & M32
    r = lowbias32(h ^ k1) >> np.uint64(8)
    return r.astype(np.float32) * np.float32(1.0 / 16777216.0)


def masks(sites, counter):
    """``sites``: [(shape, p), ...] -> the factor arrays one launch at stream counter ``counter`` writes (float32)."""
    sizes = [int(np.prod(shape)) for shape, _ in sites]
    starts, total = [], 0
    for n in sizes:
        starts.append(total)
        total += (n + 3) // 4 * 4
    u = uniforms(counter, total)
    out = []
    for (shape, p), s0, n in zip(sites, starts, sizes):
        p32 = np.float32(p)
        scale = np.float32(1.0) / (np.float32(1.0) - p32)
        out.append(np.where(u[s0:s0 + n] < p32, np.float32(0.0), scale).astype(np.float32).reshape(shape))
    return out
