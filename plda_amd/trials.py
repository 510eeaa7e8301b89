"""plda_amd/trials.py -- trial lists and the score file around the hot path (SURVEY.md section 8f
rank 1): what /root/reference/scoring/scorePLDA.py does right after `plda.transform`.

  parse_trial_ref  = test_ref   (scorePLDA.py:40-50)   "target enrol-utt" lines
  parse_mlf        = mlffile    (scorePLDA.py:55-73)   HTK master-label files
  score_trial_list = the scoring loop (scorePLDA.py:299-321), same skip/warn rules and the
                     same output bytes ("{} {}-{} {:.3f}\\n", :317-318), but ONE batched
                     fp64 trial-list launch instead of one `plda.score` call per trial.
"""
import logging
from collections import OrderedDict

import numpy as np

log = logging.getLogger(__name__)


def parse_trial_ref(path):
    """{target model: [[utt, enrol model], ...]} in file order (scorePLDA.py:40-50)."""
    tests = OrderedDict()
    with open(path, "r") as fp:
        for line in fp:
            line = line.rstrip("\n")
            if not line.strip():
                continue
            claimed, pair = line.split()[:2]
            parts = pair.split("-")
            tests.setdefault(claimed, []).append(["-".join(parts[1:]), parts[0]])
    return tests


def parse_mlf(path):
    """Same structure from an HTK MLF (scorePLDA.py:55-73)."""
    tests = OrderedDict()
    with open(path, "r") as fp:
        next(fp)  # "#!MLF!#"
        for line in fp:
            line = line.rstrip("\n")
            if line.startswith('"'):
                stem = line.split(".")[0].split("/")[1]
                parts = stem.split("-")
                claimed = next(fp).rstrip("\n")
                tests.setdefault(claimed, []).append(["-".join(parts[1:]), parts[0]])
    return tests


def score_trial_list(plda, trial_ref, enrol_vecs, test_vecs, model_to_spk, utt_to_spk,
                     out, znorm=True):
    """Score every (model, utterance) trial of `trial_ref` and write the reference's score
    file.  Returns (n_scored, n_errors).  `plda` is a liblda.PLDA / plda_amd.MPlda."""
    enrol_ids = list(enrol_vecs.keys())
    test_ids = list(test_vecs.keys())
    epos = {k: i for i, k in enumerate(enrol_ids)}
    tpos = {k: i for i, k in enumerate(test_ids)}
    e_idx, t_idx, rows = [], [], []
    errors = 0
    for model, vals in trial_ref.items():
        if model not in model_to_spk:                       # scorePLDA.py:303-306
            errors += 1
            log.warning("Enrolemodel %s not found in the labels", model)
            continue
        spk = model_to_spk[model]
        for utt, claimed in vals:
            if utt not in utt_to_spk:                        # :309-312
                log.warning("Utterance %s not found in the testset", utt)
                errors += 1
                continue
            e_idx.append(epos[spk])
            t_idx.append(tpos[utt_to_spk[utt]])
            rows.append((model, claimed, utt))
    if rows:
        scorer = getattr(plda, "score_trials", None) or plda._instance.score_trials
        scores = scorer(enrol_vecs, test_vecs, np.asarray(e_idx), np.asarray(t_idx), znorm)
        for (model, claimed, utt), s in zip(rows, scores):
            out.write("{} {}-{} {:.3f}\n".format(model, claimed, utt, s))   # :317-318
    if errors > 0:
        log.warning("Overall %i errors occured during the testing phase!", errors)
    return len(rows), errors


def score_trial_list_lda(lda, trial_ref, model_to_class, utt_to_vec, out, chunk=8192):
    """The scoring loop of scoring/scoreLDA.py:228-248 in batched form.

    The reference calls `lda.predict_log_proba(testdvector[np.newaxis, :])[0]` once per trial
    (:240-241) and keeps the entry of the enrol model's speaker (:244); here every distinct test
    utterance goes through ONE predict_log_proba per chunk of utterances (a [chunk, K] GEMM +
    log-softmax on the GPU) and the trial's entry is gathered.  Output lines are byte-identical:
    "{} {}-{} {:.3f}\\n".format(enrolemodel, targetmdl, testutt, finalscore) (:245-246).
    `lda`: liblda.LDA fitted on the speaker-numbered d-vectors; `model_to_class`: {speaker: class index}
    (scoreLDA.py:215-217); `utt_to_vec`: {utterance: d-vector}.  Returns (n_scored, n_errors)."""
    rows, utts, upos = [], [], {}
    errors = 0
    for model, vals in trial_ref.items():
        if model not in model_to_class:                            # scoreLDA.py:229-232
            errors += 1
            log.warning("Enrolemodel %s not found in the labels", model)
            continue
        cls = model_to_class[model]
        for utt, claimed in vals:
            if utt not in utt_to_vec:                       # :235-238
                log.warning("Utterance %s not found in the testset", utt)
                errors += 1
                continue
            if utt not in upos:
                upos[utt] = len(utts)
                utts.append(utt)
            rows.append((model, claimed, utt, cls, upos[utt]))
    if rows:
        picked = {}
        by_utt = {}
        for r, (_, _, _, spk, u) in enumerate(rows):
            by_utt.setdefault(u, []).append((r, spk))
        for c0 in range(0, len(utts), chunk):
            feats = np.stack([np.asarray(utt_to_vec[u], dtype=np.float64) for u in utts[c0:c0 + chunk]])
            lp = lda.predict_log_proba(feats)
            for u in range(c0, min(len(utts), c0 + chunk)):
                for r, spk in by_utt.get(u, ()):
                    picked[r] = lp[u - c0, spk]
        for r, (model, claimed, utt, _, _) in enumerate(rows):
            out.write("{} {}-{} {:.3f}\n".format(model, claimed, utt, picked[r]))
    if errors > 0:
        log.warning("Overall %i happened while processing the testutterances. The scores may not be complete", errors)
    return len(rows), errors
