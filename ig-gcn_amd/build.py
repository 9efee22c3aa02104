"""Build libigcn.so (hipcc, gfx950 only) in-tree: ig-gcn_amd/lib/libigcn.so.

Usage:  python ig-gcn_amd/build.py [--force]
hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libigcn.so")
SOURCES = ["plan.hip", "sgcn.hip", "sgcn_fused.hip", "sgcn_dense.hip", "go.hip", "readout.hip", "loss.hip", "attn_core.hip", "attn_mfma.hip", "attn_bf16.hip", "attn_split.hip", "gemm.hip", "proj.hip", "head.hip", "misc.hip", "gdc.hip", "comm.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-result", "-Werror=return-type"]
FLAGS += os.environ.get("IGCN_HIPCC_EXTRA", "").split()       # e.g. -DGO_ABL_PROBE for tools/go_probe.py


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in sorted(os.listdir(root)):
            if f.endswith((".hip", ".h")):
                with open(os.path.join(root, f), "rb") as fh:
                    h.update(f.encode() + fh.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libigcn.digest")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = _hipcc()
    objs = []

    def cc(src):
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    for o in objs:
        os.remove(o)
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
