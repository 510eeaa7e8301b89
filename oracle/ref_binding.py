"""oracle/ref_binding.py -- ctypes over oracle/_ref/libchtk_ref.so: the REFERENCE's own HTK reader
(/root/reference/chtk/chtk.cpp, compiled where it lies by `make -C oracle ref`).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_ref", "libchtk_ref.so")
REFERENCE = os.environ.get("PLDA_REFERENCE", "/root/reference")
_lib = None


def available():
    return os.path.exists(SO) or os.path.exists(os.path.join(REFERENCE, "chtk", "chtk.cpp"))


def build():
    """Compile the reference's chtk.cpp (only possible where /root/reference exists)."""
    if os.path.exists(os.path.join(REFERENCE, "chtk", "chtk.cpp")):
        subprocess.check_call(["make", "-C", HERE, "ref", "REFERENCE=" + REFERENCE], stdout=subprocess.DEVNULL)
    return os.path.exists(SO)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO) and not build():
            raise RuntimeError("oracle/_ref/libchtk_ref.so is missing and the reference source is not here")
        _lib = C.CDLL(SO)
        _lib.ref_htk_header.restype = C.c_int
        _lib.ref_htk_header.argtypes = [C.c_char_p, C.POINTER(C.c_longlong)]
        _lib.ref_htk_load.restype = C.c_longlong
        _lib.ref_htk_load.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_longlong]
    return _lib


def header(path):
    out = (C.c_longlong * 4)()
    if lib().ref_htk_header(path.encode(), out):
        raise RuntimeError("reference load_header failed")
    return tuple(int(v) for v in out)


def load(path, frm_ext=0):
    """uint32 bits [nsamples, words] of what chtk::htk_load returns."""
    n = lib().ref_htk_load(path.encode(), int(frm_ext), None, 0)
    if n < 0:
        raise RuntimeError("reference htk_load failed")
    buf = np.zeros(max(n, 1), np.uint8)
    if n and lib().ref_htk_load(path.encode(), int(frm_ext), buf.ctypes.data, n) != n:
        raise RuntimeError("reference htk_load failed")
    return buf[:n].view(np.uint32)
