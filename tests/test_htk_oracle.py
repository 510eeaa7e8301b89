"""CPU: the HTK oracle (oracle/htk_oracle_np.py) against the REFERENCE's own reader -- live through
oracle/_ref/libchtk_ref.so (the reference's chtk.cpp compiled where it lies) when it is available, and
against the golden vectors recorded from it (tests/golden/htk_cases.npz) always.  Bit-exact."""
import os

import numpy as np
import pytest

from oracle import htk_oracle_np as ho, ref_binding as rb

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "htk_cases.npz"))
CASES = sorted(k[:-5] for k in GOLD.files if k.endswith("_file"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    raw = GOLD[name + "_file"].tobytes()
    assert tuple(GOLD[name + "_header"]) == ho.load_header(raw)
    got = ho.htk_load(raw, int(GOLD[name + "_frm_ext"]))
    np.testing.assert_array_equal(got.ravel(), GOLD[name + "_out"])


@pytest.mark.skipif(not rb.available(), reason="neither oracle/_ref/libchtk_ref.so nor the reference source is here")
@pytest.mark.parametrize("n,dim,f", [(0, 7, 2), (1, 1, 0), (3, 2, 5), (33, 24, 2), (120, 39, 0), (64, 64, 4)])
def test_oracle_matches_live_reference(tmp_path, n, dim, f):
    rng = np.random.default_rng(n * 131 + dim * 7 + f)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    p = str(tmp_path / "x.htk")
    ho.write_htk(p, x, sample_period=12345, parmkind=70)
    raw = open(p, "rb").read()
    assert rb.header(p) == ho.load_header(raw)
    np.testing.assert_array_equal(ho.htk_load(raw, f).ravel(), rb.load(p, f))
    if f == 0:                                       # decode(encode(x)) == x, bit for bit
        np.testing.assert_array_equal(ho.htk_load(raw, 0).view(np.float32), x)


@pytest.mark.skipif(not rb.available(), reason="reference reader not available")
def test_negative_header_fields_follow_the_reference(tmp_path):
    p = str(tmp_path / "neg.htk")
    open(p, "wb").write(bytes([0xff, 0xff, 0xff, 0xfe, 0x80, 0, 0, 1, 0xff, 0xfc, 0x80, 0x01]))
    assert rb.header(p) == ho.load_header(open(p, "rb").read()) == (-2, -(2 ** 31) + 1, -4, -32767)


def test_bad_frame_size_is_rejected():
    with pytest.raises(ValueError):
        ho.htk_load(bytes([0, 0, 0, 1, 0, 0, 0, 1, 0, 6, 0, 9]) + b"\0" * 6, 0)
