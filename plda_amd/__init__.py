"""plda_amd -- MI355X (gfx950) native PLDA engine behind the `liblda.PLDA` API.

Layout (only what the hot path needs):
  csrc/       hand-written HIP kernels + the C ABI (include/plda_hip.h)
  lib/        built libplda_hip.so (in-tree, git-ignored)
  _native.py  ctypes binding of the C ABI (fails loudly if the .so or a GPU is missing)
  libplda.py  `MPlda`: counterpart of the reference's CPython type libplda.MPlda
  sharding.py row-sharded trials matrix across ranks (torch.distributed / RCCL)
"""
from .libplda import MPlda  # noqa: F401

__all__ = ["MPlda"]
