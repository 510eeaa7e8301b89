"""CPU: trial-list parsing and score-file bytes (reference: scoring/scorePLDA.py:40-73,299-321)."""
import io

import numpy as np
import pytest

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


def test_kaldi_plda_file_round_trip_and_layout(tmp_path):
    """plda_amd/kaldi_io.py: binary and text Kaldi `Plda` files round-trip, and the binary layout is the one
    Kaldi's WriteToken / WriteBasicType / Vector::Write / Matrix::Write conventions give (format restated;
    unpinned -- there is no Kaldi build to check against)."""
    import struct
    from plda_amd import kaldi_io
    rng = np.random.default_rng(4)
    mean, T, psi = rng.standard_normal(5), rng.standard_normal((3, 5)), np.sort(rng.random(3))[::-1].copy()
    for binary in (True, False):
        p = str(tmp_path / ("plda_%d" % binary))
        kaldi_io.write_plda(p, mean, T, psi, binary)
        m2, t2, p2 = kaldi_io.read_plda(p)
        assert np.array_equal(m2, mean) and np.array_equal(t2, T) and np.array_equal(p2, psi)
    raw = open(str(tmp_path / "plda_1"), "rb").read()
    want = (b"\0B<Plda> DV \x04" + struct.pack("<i", 5) + mean.astype("<f8").tobytes() +
            b"DM \x04" + struct.pack("<i", 3) + b"\x04" + struct.pack("<i", 5) + T.astype("<f8").tobytes() +
            b"DV \x04" + struct.pack("<i", 3) + psi.astype("<f8").tobytes() + b"</Plda> ")
    assert raw == want
    # a float32 file (tokens FV / FM) is accepted on reading
    f32 = (b"\0B<Plda> FV \x04" + struct.pack("<i", 2) + np.array([1, 2], "<f4").tobytes() +
           b"FM \x04" + struct.pack("<i", 1) + b"\x04" + struct.pack("<i", 2) + np.array([[3, 4]], "<f4").tobytes() +
           b"FV \x04" + struct.pack("<i", 1) + np.array([0.5], "<f4").tobytes() + b"</Plda> ")
    p = str(tmp_path / "f32")
    open(p, "wb").write(f32)
    m3, t3, p3 = kaldi_io.read_plda(p)
    assert m3.tolist() == [1.0, 2.0] and t3.tolist() == [[3.0, 4.0]] and p3.tolist() == [0.5]
    # Kaldi's text layout (matrix rows on their own lines, last row closed by " ]")
    txt = "<Plda>  [ 1 2 ]\n [\n  3 4 \n  5 6 ]\n [ 0.5 0.25 ]\n</Plda> "
    p = str(tmp_path / "txt")
    open(p, "w").write(txt)
    m4, t4, p4 = kaldi_io.read_plda(p)
    assert m4.tolist() == [1.0, 2.0] and t4.tolist() == [[3.0, 4.0], [5.0, 6.0]] and p4.tolist() == [0.5, 0.25]
    open(p, "w").write("<Nnet> ")
    with pytest.raises(ValueError):
        kaldi_io.read_plda(p)
