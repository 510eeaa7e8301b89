"""liblda/plda.py -- the reference's `PLDA` class (python/liblda/plda.py:4-51): the same
four one-line delegations, onto plda_amd.MPlda (HIP kernels on MI355X)."""
from plda_amd.libplda import MPlda


class PLDA(object):

    def __init__(self, device=0):
        self._instance = MPlda(device)

    def fit(self, x, y, iters=10):
        """Fit the model on background data x (nsamples, featdim), uint labels y
        (plda.py:9-10)."""
        return self._instance.fit(x, y, iters)

    def transform(self, x, y):
        """Transform vectors x with labels y into the PLDA space: {label: (n, vector)}
        (plda.py:12-23)."""
        return self._instance.transform(x, y)

    def norm(self, vectors, transformedvecs, numutts=0):
        """Estimate z-norm mean/std of the enrol models `transformedvecs` against the
        held-out `vectors` (plda.py:25-37)."""
        return self._instance.norm(vectors, transformedvecs, numutts)

    def score(self, target, xvec, yvec):
        """Score enrol model xvec=(n, vec) of id `target` against test yvec=(n, vec)
        (plda.py:39-51).  Returns a float."""
        return self._instance.score(target, xvec, yvec)

    # ---- batched extensions (what the reference's callers loop over in Python) ----
    def score_matrix(self, enrol, test, znorm=True):
        """float32 [M, Nt] matrix of score(id_i, enrol_i, test_j) in one launch."""
        return self._instance.score_matrix(enrol, test, znorm)

    def score_trials(self, enrol, test, e_idx, t_idx, znorm=True):
        return self._instance.score_trials(enrol, test, e_idx, t_idx, znorm)

    def transform_array(self, xbar, num_examples=1):
        return self._instance.transform_array(xbar, num_examples)

    def save(self, path):
        return self._instance.save(path)

    def load(self, path):
        return self._instance.load(path)

    def save_kaldi(self, path, binary=True):
        """Write the model in Kaldi's `Plda` file layout (plda_amd/kaldi_io.py)."""
        return self._instance.save_kaldi(path, binary)

    def load_kaldi(self, path):
        self._instance.load_kaldi(path)
        return self
