"""MI355X-native GenCast denoiser + DPM-Solver++2S sampling path.

Python host code -> ctypes -> `libgencast_hip.so` (C ABI, include/gencast_hip.h)
-> hand-written HIP kernels for gfx950.  No PyTorch / JAX / Triton in here.
Import as `gencast_flax_nnx_amd` (alias module at the repo root).
"""
from . import geometry  # noqa: F401

__all__ = ["geometry"]
