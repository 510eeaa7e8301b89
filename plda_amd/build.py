"""plda_amd/build.py -- builds plda_amd/lib/libplda_hip.so for gfx950 with hipcc.

hipcc cross-compiles without a GPU; the .so is kept in-tree (git-ignored) so it
travels to the GPU box with the snapshot.

`--diag` (build(diag=True)) builds the DIAGNOSTIC library plda_amd/lib/libplda_hip_diag.so from the same sources with
-DPLDA_DIAG=1 (objects under lib/diag/): it additionally contains the measurement arms of the trials GEMM (bounding arms
that return garbage scores, clock-stamp and stage-depth arms; csrc/common.hpp).  The profiling scripts and bench.py's
shader-clock reading load it (plda_amd._native.load(diag=True) / PLDA_LIB_DIAG=1); nothing else does.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libplda_hip.so")
SO_DIAG = os.path.join(LIBDIR, "libplda_hip_diag.so")
SOURCES = ["api.hip", "score.hip", "linalg.hip", "fit.hip", "frontend.hip", "eer.hip", "lda.hip", "comm.hip", "eig_dc.hip", "hostio.hip", "transform.hip"]
HEADERS = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "hostio.hpp"), os.path.join(CSRC, "sweep_mfma.inc"), os.path.join(CSRC, "score_bt4.inc"), os.path.join(CSRC, "score_bf16x3.inc"), os.path.join(CSRC, "syrk_blk.inc"), os.path.join(HERE, "..", "include", "plda_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


# score.hip: the one-lane queue atomic of trials_gemm_bt4_kernel must stay ONE asynchronous instruction -- the atomic optimizer
# rewrites it into a wave reduction that reads the result back on the spot (a memory round trip in the MFMA stream); the
# tile fetch's two values are deliberately defined on one path only (score_bt4.inc: the warning is silenced by a pragma around
# that include, not for the file)
EXTRA = {"score.hip": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, diag=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(LIBDIR, "diag") if diag else LIBDIR
    so = SO_DIAG if diag else SO
    flags = FLAGS + (["-DPLDA_DIAG=1"] if diag else [])
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc] + flags + EXTRA.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (src, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    if force or procs or _stale(so, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + ["-ldl", "-lpthread"]   # librccl is opened lazily by plda_comm_init (csrc/comm.hip)
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
