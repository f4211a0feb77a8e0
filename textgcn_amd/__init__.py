"""MI355X-native LightGCN propagation + scoring path (drop-in for TextGCN's BaseModel hot path).

Kernels live in csrc/ (HIP, gfx950) behind the C ABI of include/tgcn.h; this package is the host-side
mirror of the reference's Python surface.  There is no CPU fallback: without libtgcn.so and a ROCm GPU the
compute entry points raise.
"""
from .graph import NormGraph, split_plan_arrays, train_mask_csr  # noqa: F401

__all__ = ['NormGraph', 'split_plan_arrays', 'train_mask_csr']
