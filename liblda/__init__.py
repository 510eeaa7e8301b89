"""liblda -- drop-in for the reference's Python package (python/liblda/__init__.py:1-3).

`from liblda import PLDA` works as in RicherMans/PLDA; the object underneath is the
MI355X engine (plda_amd.MPlda) instead of the Kaldi-backed CPython-2 extension.
`LDA` (python/liblda/lda.py) runs on the same engine.
"""
from .plda import PLDA
from .lda import LDA

__all__ = ["PLDA", "LDA"]
