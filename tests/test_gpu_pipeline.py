"""End to end on the GPU, the way scoring/extractdvector.py + scoring/scorePLDA.py + scoring/eer.py chain the
pieces: HTK files of frame-level d-vectors -> per-utterance pooling -> PLDA fit / transform / norm ->
trial-list scores in the reference's file format -> EER.  The same chain is run on the oracles (HTK reader
pinned by the reference's chtk.cpp, pooling / PLDA / EER restatements) and must give the same score file
(to the 3 decimals it prints) and the same error rates."""
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_htk_to_eer_pipeline(tmp_path, oracle):
    from liblda import PLDA
    from oracle import htk_oracle_np as ho, plda_oracle_np as onp
    from plda_amd import eer as geer, htk
    from plda_amd.dvector import pool
    from plda_amd.trials import parse_trial_ref, score_trial_list
    rng = np.random.default_rng(123)
    dim, nspk = 24, 12
    centers = rng.standard_normal((nspk, dim))

    def make_utts(prefix, per_spk):
        paths, spk = [], []
        for s in range(nspk):
            for u in range(per_spk):
                frames = centers[s] + 0.8 * rng.standard_normal(dim) + 0.5 * rng.standard_normal((int(rng.integers(20, 60)), dim))
                p = str(tmp_path / ("%s_s%02d_u%d.htk" % (prefix, s, u)))
                ho.write_htk(p, frames.astype(np.float32))
                paths.append(p); spk.append(s)
        return paths, np.array(spk, np.uint64)

    bkg_p, bkg_y = make_utts("bkg", 12)
    enr_p, enr_y = make_utts("enr", 3)
    tst_p, tst_y = make_utts("tst", 4)
    held_p, _ = make_utts("held", 2)

    # ---- GPU chain ----
    plda = PLDA()
    eng = plda._instance

    def dvectors(paths):
        frames, off = htk.htk_load_batch(paths, 0, engine=eng)
        return pool(frames, off, "mean", engine=eng)
    B, E, T, H = dvectors(bkg_p), dvectors(enr_p), dvectors(tst_p), dvectors(held_p)
    plda.fit(B, bkg_y, 6)
    enrol = plda.transform(E, enr_y)                               # one model per speaker (3 utterances each)
    test = plda.transform(T, np.arange(len(tst_p), dtype=np.uint64))
    plda.norm(H, enrol)
    lines = []
    for j, ts in enumerate(tst_y):
        for s in range(nspk):
            if (j + s) % 3 == 0 or s == ts:
                lines.append("spk%02d spk%02d-utt%03d %d\n" % (s, int(ts), j, int(s == ts)))
    ref_path = tmp_path / "test_ref"
    ref_path.write_text("".join(lines))
    refs = parse_trial_ref(str(ref_path))
    spk2id = {"spk%02d" % s: s for s in range(nspk)}
    utt2id = {"utt%03d" % j: j for j in range(len(tst_p))}
    got = io.StringIO()
    n, err = score_trial_list(plda, refs, enrol, test, spk2id, utt2id, got)
    assert err == 0 and n == len(lines)

    # ---- oracle chain ----
    def dvectors_ref(paths):
        frs = [ho.htk_load(open(p, "rb").read(), 0).view(np.float32) for p in paths]
        off = np.concatenate([[0], np.cumsum([len(f) for f in frs])])
        return onp.dvector_pool(np.concatenate(frs), off, "mean")
    Br, Er, Tr, Hr = dvectors_ref(bkg_p), dvectors_ref(enr_p), dvectors_ref(tst_p), dvectors_ref(held_p)
    np.testing.assert_allclose(B, Br, rtol=1e-12, atol=1e-14)
    model = oracle.fit(Br, bkg_y, 6)
    _, ec, ev = oracle.transform_groups(model, Er, enr_y)
    _, _, tv = oracle.transform_groups(model, Tr, np.arange(len(tst_p), dtype=np.uint64))
    zm, zs = oracle.norm(model, Hr, ev)
    S = oracle.score_block(model["psi"], ev, ec, tv, zm, zs)
    want = io.StringIO()
    for enrolemodel, vals in refs.items():
        for testutt, targetmdl in vals:
            want.write("{} {}-{} {:.3f}\n".format(enrolemodel, targetmdl, testutt, S[spk2id[enrolemodel], utt2id[testutt]]))
    g_lines, w_lines = got.getvalue().splitlines(), want.getvalue().splitlines()
    assert len(g_lines) == len(w_lines)
    for a, b in zip(g_lines, w_lines):
        assert a.rsplit(" ", 1)[0] == b.rsplit(" ", 1)[0]
        assert abs(float(a.rsplit(" ", 1)[1]) - float(b.rsplit(" ", 1)[1])) <= 0.0011      # last printed digit
    # ---- EER of the written scores (scoring/eer.py reads such a file back) ----
    sc = np.array([float(l.rsplit(" ", 1)[1]) for l in g_lines], np.float32)
    lab = np.array([l.split()[0] == l.split()[1].split("-")[0] for l in g_lines])
    thr, far, frr, e = geer.eer_from_lists(eng, sc[lab], sc[~lab])
    rthr, rfar, rfrr, re = onp.eer(sc[~lab], sc[lab])
    assert (far, frr, e) == (rfar, rfrr, re) and thr == pytest.approx(rthr, rel=1e-12)
    assert e < 0.2                                                  # the synthetic speakers are separable
