"""plda_amd/htk.py -- HTK feature files on the MI355X engine: the reference's reader
chtk::htk_load / chtk::load_header (/root/reference/chtk/chtk.cpp:38-110, used by the compiled
modules at src/kaldi-utils.hpp:22) behind the same two calls, plus a batched form.

The 12-byte header is parsed here (host logic); the frames -- byte swap of every big-endian
float and the +-frm_ext context stacking with clamped edges -- are decoded by
libplda_hip.so (csrc/frontend.hip: htk_frames_kernel), bit-exact with the reference.
"""
import struct

import numpy as np

from . import _native as N
from .libplda import MPlda, _ptr


def _open(fname):
    try:
        return open(fname, "rb")
    except OSError:
        raise RuntimeError("File " + str(fname) + " cannot be opened !\n")     # chtk.cpp:44-46,101-103


def load_header(fname):
    """chtk::load_header(std::string) (chtk.cpp:97-110): dict of nsamples, sample_period,
    samplesize (bytes per frame), parmkind -- signed, as the reference's int / short fields hold them."""
    with _open(fname) as f:
        raw = f.read(12)
    raw = raw + b"\0" * (12 - len(raw))
    n, p, s, k = struct.unpack(">iihh", raw)
    return dict(nsamples=n, sample_period=p, samplesize=s, parmkind=k)


def _read(fname):
    with _open(fname) as f:
        raw = f.read()
    raw_h = raw[:12] + b"\0" * max(0, 12 - len(raw))
    n, _, size, _ = struct.unpack(">IIHH", raw_h)          # htk_load keeps them unsigned (ntohl / ntohs, :53)
    if size % 4:
        raise ValueError("HTK file %s: samplesize %d is not a multiple of 4 (the reference's frame swap "
                         "throws std::out_of_range here, chtk.cpp:60-65)" % (fname, size))
    body = raw[12:12 + n * size]
    return n, size, body


def htk_load_batch(fnames, frm_ext=0, engine=None):
    """Decode many files in one GPU call.  Returns (frames float32 [T, (2 frm_ext + 1) * dim],
    offsets int64 [U + 1]); file u owns rows offsets[u]:offsets[u+1].  All files must share the
    frame size."""
    eng = engine if engine is not None else MPlda(0)
    metas = [_read(f) for f in fnames]
    if not metas:
        return np.zeros((0, 0), np.float32), np.zeros(1, np.int64)
    size = metas[0][1]
    for f, m in zip(fnames, metas):
        if m[1] != size:
            raise ValueError("HTK batch: %s has %d bytes per frame, expected %d" % (f, m[1], size))
    counts = np.array([m[0] for m in metas], np.int64)
    frame_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    w = size // 4
    file_off = (frame_off[:-1] * w).astype(np.int64)          # data sections back to back, in 32-bit words
    blob = np.zeros(int(frame_off[-1]) * size, np.uint8)      # zero padding = what a short file reads as
    for m, fo in zip(metas, file_off):
        blob[fo * 4: fo * 4 + len(m[2])] = np.frombuffer(m[2], np.uint8)
    out = np.empty((int(frame_off[-1]), (2 * frm_ext + 1) * w), np.float32)
    if out.size:
        N.check(eng._h, eng._lib.plda_htk_frames(eng._h, _ptr(blob), blob.nbytes, _ptr(file_off), _ptr(frame_off),
                                                 len(metas), int(size), int(frm_ext), _ptr(out)))
    return out, frame_off


def htk_load(fname, frm_ext=0, engine=None):
    """chtk::htk_load(fname, FRM_EXT) (chtk.cpp:38-88): float32 [nsamples, (2 FRM_EXT + 1) * dim]."""
    return htk_load_batch([fname], frm_ext, engine)[0]
