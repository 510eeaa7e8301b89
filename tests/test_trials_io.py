"""CPU: trial-list parsing and score-file bytes (reference: scoring/scorePLDA.py:40-73,299-321)."""
import io

import numpy as np

from plda_amd import trials


class _FakePLDA:
    def score_trials(self, enrol, test, e_idx, t_idx, znorm=True):
        e = list(enrol.values()); t = list(test.values())
        return np.array([float(e[i][1][0]) * 10 + float(t[j][1][0]) for i, j in zip(e_idx, t_idx)])


def test_parse_trial_ref_and_mlf(tmp_path):
    ref = tmp_path / "trials.txt"
    ref.write_text("spkA spkA-utt-1 extra\nspkA spkB-utt2\nspkB spkA-utt-1\n\n")
    t = trials.parse_trial_ref(str(ref))
    assert list(t) == ["spkA", "spkB"]
    assert t["spkA"] == [["utt-1", "spkA"], ["utt2", "spkB"]] and t["spkB"] == [["utt-1", "spkA"]]
    mlf = tmp_path / "t.mlf"
    mlf.write_text('#!MLF!#\n"*/spkA-utt-1.lab"\nspkA\n.\n"*/spkB-utt2.lab"\nspkA\n.\n')
    assert trials.parse_mlf(str(mlf)) == {"spkA": [["utt-1", "spkA"], ["utt2", "spkB"]]}


def test_score_file_bytes_and_skip_rules():
    enrol = {0: (2, np.array([1.0])), 1: (1, np.array([2.0]))}
    test = {0: (1, np.array([0.25])), 1: (1, np.array([0.5]))}
    ref = {"spkA": [["utt-1", "spkA"], ["missing", "spkA"], ["utt2", "spkB"]], "ghost": [["utt2", "spkA"]]}
    out = io.StringIO()
    n, err = trials.score_trial_list(_FakePLDA(), ref, enrol, test, {"spkA": 0, "spkB": 1}, {"utt-1": 0, "utt2": 1}, out)
    assert (n, err) == (2, 2)
    assert out.getvalue() == "spkA spkA-utt-1 10.250\nspkA spkB-utt2 10.500\n"
