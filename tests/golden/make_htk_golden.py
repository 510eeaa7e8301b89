"""tests/golden/make_htk_golden.py -- golden vectors for the HTK reader, produced by the REFERENCE itself.

Run in the build container only:  python tests/golden/make_htk_golden.py
It writes small HTK files, runs the reference's own chtk::load_header / chtk::htk_load on them through
oracle/_ref/libchtk_ref.so (compiled by `make -C oracle ref` from /root/reference/chtk/chtk.cpp, where it
lies) and stores file bytes + the reference's outputs as data in tests/golden/htk_cases.npz.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import htk_oracle_np as ho, ref_binding as rb   # noqa: E402


def main():
    rng = np.random.default_rng(20260928)
    d = tempfile.mkdtemp()
    rec = {}
    cases = [("empty", 0, 5, 1), ("one", 1, 3, 2), ("two", 2, 4, 0), ("odd", 17, 13, 3), ("mfcc", 50, 40, 1),
             ("wide", 9, 257, 2)]
    for name, n, dim, f in cases:
        x = rng.standard_normal((n, dim)).astype(np.float32)
        if n:
            x.view(np.uint32)[0, 0] = 0x7fc01234     # a NaN payload must survive bit for bit
        p = os.path.join(d, name + ".htk")
        ho.write_htk(p, x, sample_period=1 + int(rng.integers(1, 10 ** 6)), parmkind=int(rng.integers(0, 100)))
        rec[name + "_file"] = np.frombuffer(open(p, "rb").read(), np.uint8)
        rec[name + "_frm_ext"] = np.array(f)
        rec[name + "_header"] = np.array(rb.header(p))
        rec[name + "_out"] = rb.load(p, f)
    # a file shorter than its header claims: the missing bytes read as zeros (chtk.cpp:56-57)
    p = os.path.join(d, "short.htk")
    ho.write_htk(p, rng.standard_normal((9, 6)).astype(np.float32))
    raw = open(p, "rb").read()[:-31]
    open(p, "wb").write(raw)
    rec["short_file"] = np.frombuffer(raw, np.uint8)
    rec["short_frm_ext"] = np.array(1)
    rec["short_header"] = np.array(rb.header(p))
    rec["short_out"] = rb.load(p, 1)
    np.savez_compressed(os.path.join(HERE, "htk_cases.npz"), **rec)
    print("wrote htk_cases.npz:", sorted(k[:-4] for k in rec if k.endswith("_out")))


if __name__ == "__main__":
    main()
