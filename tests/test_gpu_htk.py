"""GPU parity of the HTK reader (plda_amd.htk, csrc/frontend.hip:htk_frames_kernel) through the C ABI,
bit-exact: against the REFERENCE's own outputs (tests/golden/htk_cases.npz, recorded from chtk.cpp via
oracle/_ref), against the live reference library when it travelled with the snapshot, and against the
NumPy oracle at a large batch; then the pipeline into the d-vector pooling kernel."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "htk_cases.npz"))
CASES = sorted(k[:-5] for k in GOLD.files if k.endswith("_file"))


@pytest.mark.parametrize("name", CASES)
def test_htk_load_matches_reference_golden(tmp_path, name):
    from plda_amd import htk
    p = str(tmp_path / (name + ".htk"))
    open(p, "wb").write(GOLD[name + "_file"].tobytes())
    hd = htk.load_header(p)
    assert (hd["nsamples"], hd["sample_period"], hd["samplesize"], hd["parmkind"]) == tuple(GOLD[name + "_header"])
    out = htk.htk_load(p, int(GOLD[name + "_frm_ext"]))
    assert out.dtype == np.float32
    np.testing.assert_array_equal(out.view(np.uint32).ravel(), GOLD[name + "_out"])


def test_htk_batch_matches_oracle_and_live_reference(tmp_path):
    from oracle import htk_oracle_np as ho, ref_binding as rb
    from plda_amd import htk
    rng = np.random.default_rng(11)
    paths, raws = [], []
    for u in range(300):
        n = int(rng.integers(0, 90))
        x = rng.standard_normal((n, 40)).astype(np.float32)
        p = str(tmp_path / ("u%03d.htk" % u))
        ho.write_htk(p, x)
        if u % 50 == 7:                                  # a truncated file in the batch
            raw = open(p, "rb").read()
            open(p, "wb").write(raw[: len(raw) - 50])
        paths.append(p)
        raws.append(open(p, "rb").read())
    for f in (0, 2):
        frames, off = htk.htk_load_batch(paths, f)
        assert frames.shape == (off[-1], (2 * f + 1) * 40)
        for u in (0, 7, 57, 123, 299):
            want = ho.htk_load(raws[u], f)
            np.testing.assert_array_equal(frames[off[u]:off[u + 1]].view(np.uint32), want)
            if rb.available():
                np.testing.assert_array_equal(frames[off[u]:off[u + 1]].view(np.uint32).ravel(), rb.load(paths[u], f))
        full = np.concatenate([ho.htk_load(r, f) for r in raws])
        np.testing.assert_array_equal(frames.view(np.uint32), full)


def test_htk_to_dvector_pipeline_and_errors(tmp_path):
    """files -> frames (GPU) -> per-utterance mean d-vector (GPU), the order scoring/extractdvector.py:118,37-39
    runs it; and the reference's error behaviour for an unreadable file."""
    from oracle import htk_oracle_np as ho, plda_oracle_np as onp
    from plda_amd import htk
    from plda_amd.dvector import pool
    rng = np.random.default_rng(3)
    paths = []
    for u in range(20):
        p = str(tmp_path / ("d%02d.htk" % u))
        ho.write_htk(p, rng.standard_normal((int(rng.integers(5, 60)), 64)).astype(np.float32))
        paths.append(p)
    frames, off = htk.htk_load_batch(paths)
    got = pool(frames, off, "mean")
    want = onp.dvector_pool(frames, off, "mean")
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-14)
    with pytest.raises(RuntimeError, match="cannot be opened"):
        htk.htk_load(str(tmp_path / "missing.htk"))
    bad = str(tmp_path / "bad.htk")
    open(bad, "wb").write(bytes([0, 0, 0, 1, 0, 0, 0, 1, 0, 6, 0, 9]) + b"\0" * 6)
    with pytest.raises(ValueError, match="not a multiple of 4"):
        htk.htk_load(bad)


def test_htk_large_batch_round_trip():
    """5M frames x 40 dims decoded from device-resident bytes: decode(encode(x)) == x, and the stacked
    context obeys the clamped-gather definition on random probes."""
    import torch
    from plda_amd import MPlda
    dev = torch.device("cuda:0")
    eng = MPlda(0)
    U, dim, F = 20000, 40, 2
    g = torch.Generator(device="cpu").manual_seed(5)
    counts = torch.randint(100, 400, (U,), generator=g)
    off = torch.zeros(U + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(counts, 0)
    T = int(off[-1])
    x = torch.randn((T, dim), dtype=torch.float32, generator=g)
    be = x.numpy().astype(">f4").view(np.uint8).reshape(-1)               # big-endian file bodies, back to back
    blob = torch.from_numpy(be.copy()).to(dev)
    file_off = (off[:-1] * dim).to(dev)
    doff = off.to(dev)
    out0 = torch.empty((T, dim), dtype=torch.float32, device=dev)
    eng._ck(eng._lib.plda_htk_frames_dev(eng._h, blob.data_ptr(), file_off.data_ptr(), doff.data_ptr(), U, T, dim * 4, 0,
                                         out0.data_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(out0.cpu().view(torch.int32), x.view(torch.int32))
    outF = torch.empty((T, (2 * F + 1) * dim), dtype=torch.float32, device=dev)
    eng._ck(eng._lib.plda_htk_frames_dev(eng._h, blob.data_ptr(), file_off.data_ptr(), doff.data_ptr(), U, T, dim * 4, F,
                                         outF.data_ptr()))
    torch.cuda.synchronize()
    rng = np.random.default_rng(0)
    for u in rng.integers(0, U, 50):
        a, b = int(off[u]), int(off[u + 1])
        i = int(rng.integers(0, b - a))
        idx = np.clip(np.arange(i - F, i + F + 1), 0, b - a - 1) + a
        want = x[idx].reshape(-1)
        assert torch.equal(outF[a + i].cpu().view(torch.int32), want.view(torch.int32))


def test_htk_unaligned_file_placement_and_empty_files():
    """Files whose data section is not 16-byte aligned inside the blob take the word path inside the same
    launch; empty files in the middle of a batch are stepped over."""
    import torch
    from oracle import htk_oracle_np as ho
    from plda_amd import MPlda
    import struct
    dev = torch.device("cuda:0")
    eng = MPlda(0)
    rng = np.random.default_rng(9)
    dim, F = 8, 1
    counts = [5, 0, 0, 33, 1, 0, 12]
    pads = [0, 0, 0, 1, 3, 0, 2]                       # words of junk in front of each file's data
    words, file_off, want = [], [], []
    pos = 0
    for n, pad in zip(counts, pads):
        words.append(rng.integers(0, 2 ** 32, pad, dtype=np.uint32)); pos += pad
        x = rng.standard_normal((n, dim)).astype(np.float32)
        body = x.astype(">f4").view(np.uint32).reshape(-1)
        file_off.append(pos); words.append(body); pos += body.size
        raw = struct.pack(">IIHH", n, 1, dim * 4, 9) + body.tobytes()
        want.append(ho.htk_load(raw, F))
    blob = torch.from_numpy(np.concatenate(words).astype(np.uint32).view(np.int32)).to(dev)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    T = int(off[-1])
    out = torch.empty((T, 3 * dim), dtype=torch.float32, device=dev)
    dfo = torch.tensor(file_off, dtype=torch.int64, device=dev)
    doff = torch.from_numpy(off).to(dev)
    eng._ck(eng._lib.plda_htk_frames_dev(eng._h, blob.data_ptr(), dfo.data_ptr(), doff.data_ptr(), len(counts), T, dim * 4, F,
                                         out.data_ptr()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), np.concatenate(want))
