"""oracle/htk_oracle_np.py -- NumPy restatement of the reference's HTK feature-file reader.

TEST INFRASTRUCTURE ONLY.  PARITY PINNED: the reference's reader (/root/reference/chtk/chtk.cpp)
is one C++11 file that compiles here, so oracle/Makefile builds it from the reference's own source
into oracle/_ref/libchtk_ref.so and tests/test_htk_oracle.py checks this file against it bit for
bit (plus tests/golden/htk_cases.npz, recorded from that library, for machines without it).

  header      chtk.cpp:90-110  12 bytes, big-endian: int32 nsamples, int32 sample_period,
                               int16 samplesize (bytes per frame), int16 parmkind
  htk_load    chtk.cpp:38-88   frames of `samplesize` bytes, every 4-byte group byte-swapped;
                               bytes missing at the end of a short file read as zeros (:56-57:
                               the buffer is zero-initialised and the failed read leaves it);
                               then frame i becomes the concatenation of frames
                               clamp(i - F .. i + F, 0, n - 1) (:71-86)
"""
import struct

import numpy as np


def parse_header(raw):
    """(nsamples, sample_period, samplesize, parmkind) as htk_load sees them (ntohl / ntohs are unsigned)."""
    if len(raw) < 12:
        raw = raw + b"\0" * (12 - len(raw))
    return struct.unpack(">IIHH", raw[:12])


def load_header(raw):
    """chtk::load_header(std::string): the same fields stored back into int / short (signed)."""
    n, p, s, k = parse_header(raw)
    to_i32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v
    to_i16 = lambda v: v - (1 << 16) if v >= (1 << 15) else v
    return to_i32(n), to_i32(p), to_i16(s), to_i16(k)


def htk_load(raw, frm_ext=0):
    """uint32 array [nsamples, (2 frm_ext + 1) * samplesize / 4] holding the IEEE bits of the
    little-endian floats the reference returns (compare bits, not values: NaN payloads survive)."""
    n, _, size, _ = parse_header(raw)
    if size % 4:
        raise ValueError("samplesize %d is not a multiple of 4" % size)   # the reference's .at() throws
    w = size // 4
    body = raw[12:12 + n * size]
    body = body + b"\0" * (n * size - len(body))
    frames = np.frombuffer(body, dtype=">u4").astype(np.uint32).reshape(n, w)
    if n == 0:
        return np.zeros((0, (2 * frm_ext + 1) * w), np.uint32)
    idx = np.clip(np.arange(n)[:, None] + np.arange(-frm_ext, frm_ext + 1)[None, :], 0, n - 1)
    return frames[idx].reshape(n, (2 * frm_ext + 1) * w)


def write_htk(path, data, sample_period=100000, parmkind=9):
    """Write float32 frames [n, dim] as an HTK file (for tests)."""
    data = np.asarray(data, np.float32)
    with open(path, "wb") as f:
        f.write(struct.pack(">IIHH", data.shape[0], sample_period, data.shape[1] * 4, parmkind))
        f.write(data.astype(">f4").tobytes())
