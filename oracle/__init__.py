"""oracle/ — TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU restatement (plain PyTorch, fp32 or fp64, no custom kernels) of the IG-GCN hot path:
the SGCN message passing over brain-ROI graphs + the GO-hierarchical attention network
over SNPs + one optimisation step.  It is the checker for the HIP path in ``ig-gcn_amd/``.

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  Nothing under ``ig-gcn_amd/`` imports it; the product path raises when
``libigcn.so`` is missing instead of falling back to anything here.

Parity pinning status:

* ``go_network``  — pinned against the reference's own ``kernel/go_model.py`` executed in the
  build container (tests/golden/make_golden.py; the only substitution is the un-vendored
  ``torch_scatter.scatter`` → ``index_add_``), fixtures in tests/golden/go_*.npz.
* ``sgcn_img_snp`` glue (masks, fusion, MHA wiring, heads, losses, train step) — pinned against
  the reference's ``kernel/sgcn_img_snp.py`` executed with PyG's ``GCNConv``/``to_dense_batch``
  replaced by ``oracle.pyg_ops`` (PyG 2.0.2 is not installable here).
* ``dropout`` — the product's own dropout-mask generator restated in numpy (the reference draws from torch's global
  generator: nothing of its numbers to pin); pinned against the kernel bit for bit on the GPU, known answers on the CPU.
* ``sgcn`` (the image-only sibling ``SGCN_GCN`` and its train loss) — pinned against the reference's
  ``kernel/sgcn.py`` executed the same way (tests/golden/sgcn_only.npz).
* ``gdc`` (PPR diffusion, top-k, column normalisation, COO emission) — pinned against the reference's own
  ``util_gdc.py`` functions, numpy float64, bit-identical outputs (tests/golden/gdc.npz).
* ``pyg_ops`` (``gcn_norm``/``GCNConv``/``to_dense_batch``/``scatter``) — third-party code that is
  absent from /root/reference (pyg=2.0.2, pytorch-scatter=2.0.9, environment.yml:183,211):
  **parity unpinned by the reference**; pinned here only by fp64 dense known-answer tests of the
  published formula  out = D^-1/2 (A_w^T + I') D^-1/2 X W^T + b  (tests/test_oracle_pyg_ops.py).

Layout: every function cites the reference file:line it follows.  The oracle is written in a
functional style over a flat ``state_dict`` (reference key names), so the very same weights
drive the reference capture, the oracle and the HIP model.
"""
