"""liblda -- drop-in for the reference's Python package (python/liblda/__init__.py:1-3).

`from liblda import PLDA` works as in RicherMans/PLDA; the object underneath is the
MI355X engine (plda_amd.MPlda) instead of the Kaldi-backed CPython-2 extension.
The reference also exports `LDA` (a NumPy/SciPy class, CPU only); it is outside the
accelerated path (out of scope) and is not provided here.
"""
from .plda import PLDA

__all__ = ["PLDA"]
