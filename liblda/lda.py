"""liblda/lda.py -- the reference's `LDA` class (python/liblda/lda.py:87-338), running on the MI355X
engine: see plda_amd/lda.py."""
from plda_amd.lda import LDA

__all__ = ["LDA"]
