"""igcn_amd — MI355X-native implementation of IG-GCN's hot path (SGCN over brain-ROI graphs fused
with the GO-hierarchical attention network over SNPs).  Python host code on PyTorch-ROCm calling
hand-written gfx950 HIP kernels through the C ABI declared in include/igcn.h (libigcn.so).

Importing the package never needs the GPU; every compute op raises if libigcn.so is missing.
"""
__version__ = "0.1.0"
