"""plda_amd/kaldi_io.py -- read / write a PLDA model in Kaldi's `Plda` file format (SURVEY.md section 8f rank 2:
"optionally Kaldi Plda::Write binary layout for interchange with Kaldi tools").

The reference keeps its model only in memory (the `Plda plda` member, pldamodule.cpp:29) and cannot
persist it; Kaldi's own tools (ivector-compute-plda, ivector-plda-scoring) exchange it as a file.  With
these two functions a model estimated by Kaldi can be scored on the GPU here, and vice versa.

FORMAT RESTATED, NOT PINNED: Kaldi is absent from the reference tree and from this image, so the layout
below is restated from Kaldi's published I/O conventions (kaldi-asr/kaldi, src/ivector/plda.cc
`Plda::Write`, src/matrix/kaldi-vector.cc / kaldi-matrix.cc `Write`, src/base/io-funcs.h) and checked only
by round trips and hand-assembled byte strings (tests/test_trials_io.py), never against a Kaldi build:
  binary:  "\\0B" "<Plda> " DV(mean) DM(transform) DV(psi) "</Plda> "
           DV = "DV " + int32(dim)            + dim float64 (little endian)
           DM = "DM " + int32(rows) int32(cols) + rows*cols float64, row-major
           int32(x) = byte 0x04 followed by the 4 little-endian bytes (WriteBasicType)
           (float32 objects, tokens "FV" / "FM", are accepted on reading)
  text:    "<Plda>  [ m0 m1 ... ]\\n [\\n  t00 t01 ...\\n  t10 ... ]\\n [ p0 p1 ... ]\\n</Plda> "
`offset` is not stored: Plda::Read recomputes it as -transform . mean (ComputeDerivedVars), as
MPlda.set_model does.
"""
import struct

import numpy as np


def _wint(v):
    return b"\x04" + struct.pack("<i", int(v))


def write_plda(path, mean, transform, psi, binary=True):
    mean = np.ascontiguousarray(mean, np.float64).reshape(-1)
    psi = np.ascontiguousarray(psi, np.float64).reshape(-1)
    transform = np.ascontiguousarray(transform, np.float64)
    if transform.ndim != 2 or transform.shape[1] != mean.shape[0] or transform.shape[0] != psi.shape[0]:
        raise ValueError("write_plda: transform must be [len(psi), len(mean)]")
    with open(path, "wb") as f:
        if binary:
            f.write(b"\0B<Plda> ")
            f.write(b"DV " + _wint(mean.shape[0]) + mean.astype("<f8").tobytes())
            f.write(b"DM " + _wint(transform.shape[0]) + _wint(transform.shape[1]) + transform.astype("<f8").tobytes())
            f.write(b"DV " + _wint(psi.shape[0]) + psi.astype("<f8").tobytes())
            f.write(b"</Plda> ")
        else:
            fmt = lambda v: " ".join(repr(float(x)) for x in v)
            out = "<Plda>  [ " + fmt(mean) + " ]\n [\n"
            rows = ["  " + fmt(r) for r in transform]
            out += " \n".join(rows) + " ]\n"
            out += " [ " + fmt(psi) + " ]\n</Plda> "
            f.write(out.encode("ascii"))


class _Bin(object):
    def __init__(self, raw, pos):
        self.raw, self.pos = raw, pos

    def token(self):
        end = self.raw.index(b" ", self.pos)
        tok = self.raw[self.pos:end].decode("ascii")
        self.pos = end + 1
        return tok

    def int32(self):
        if self.raw[self.pos] != 4:
            raise ValueError("Kaldi binary: expected a 4-byte integer at offset %d" % self.pos)
        v = struct.unpack_from("<i", self.raw, self.pos + 1)[0]
        self.pos += 5
        return v

    def floats(self, n, single):
        dt, size = ("<f4", 4) if single else ("<f8", 8)
        a = np.frombuffer(self.raw, dt, n, self.pos).astype(np.float64)
        self.pos += n * size
        return a

    def vector(self):
        tok = self.token()
        if tok not in ("DV", "FV"):
            raise ValueError("Kaldi binary: expected a vector, found %r" % tok)
        return self.floats(self.int32(), tok == "FV")

    def matrix(self):
        tok = self.token()
        if tok not in ("DM", "FM"):
            raise ValueError("Kaldi binary: expected a matrix, found %r (compressed matrices are not supported)" % tok)
        r, c = self.int32(), self.int32()
        return self.floats(r * c, tok == "FM").reshape(r, c)


def _read_text(txt):
    toks = txt.replace("[", " [ ").replace("]", " ] ").split("\n")
    flat = " \n ".join(toks).split(" ")
    flat = [t for t in flat if t != ""]
    pos = [0]

    def expect(t):
        while flat[pos[0]] == "\n":
            pos[0] += 1
        if flat[pos[0]] != t:
            raise ValueError("Kaldi text: expected %r, found %r" % (t, flat[pos[0]]))
        pos[0] += 1

    def bracket():
        expect("[")
        rows, cur = [], []
        while flat[pos[0]] != "]":
            t = flat[pos[0]]
            pos[0] += 1
            if t == "\n":
                if cur:
                    rows.append(cur)
                    cur = []
            else:
                cur.append(float(t))
        pos[0] += 1
        if cur:
            rows.append(cur)
        return rows

    expect("<Plda>")
    mean = np.array(sum(bracket(), []), np.float64)
    transform = np.array(bracket(), np.float64)
    psi = np.array(sum(bracket(), []), np.float64)
    expect("</Plda>")
    return mean, transform, psi


def read_plda(path):
    """-> (mean [Din], transform [Dout, Din], psi [Dout]) from a binary or text Kaldi Plda file."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:2] == b"\0B":
        b = _Bin(raw, 2)
        if b.token() != "<Plda>":
            raise ValueError("not a Kaldi Plda file: %s" % path)
        mean, transform, psi = b.vector(), b.matrix(), b.vector()
        if b.token() != "</Plda>":
            raise ValueError("Kaldi Plda file %s: missing </Plda>" % path)
    else:
        mean, transform, psi = _read_text(raw.decode("ascii"))
    if transform.ndim != 2 or transform.shape[1] != mean.shape[0] or transform.shape[0] != psi.shape[0]:
        raise ValueError("Kaldi Plda file %s: inconsistent dimensions" % path)
    return mean, transform, psi
