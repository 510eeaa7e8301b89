"""What a three-term bf16 split of the fp32 operands would cost in accuracy (NOT a product path).
a = a0 + a1 + a2 with a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1); the contraction keeps the six products of
order <= 2 and accumulates them in fp32 (the MFMA accumulator), against the fp32 contraction the product kernel runs
and the fp64 value.  Operands as the scoring prep makes them: A = c u / var (fp32), B = v (fp32), D = 200."""
import numpy as np


def bf16(x):
    """round-to-nearest-even of float32 to bfloat16, returned as float32"""
    b = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
    return b.astype(np.uint32).view(np.float32)


def split3(x):
    x = x.astype(np.float32)
    x0 = bf16(x)
    r = x - x0
    x1 = bf16(r)
    x2 = bf16(r - x1)
    return x0, x1, x2


rng = np.random.default_rng(0)
D, M, N = 200, 512, 512
psi = np.sort(rng.random(D) * 4 + 0.05)[::-1]
c = psi / (psi + 1.0)
var = 1.0 + psi / (psi + 1.0)
U, V = rng.standard_normal((M, D)), rng.standard_normal((N, D))
A64, B64 = c * U / var, V
A32, B32 = A64.astype(np.float32), B64.astype(np.float32)
ref = A64 @ B64.T
mag = np.abs(A64) @ np.abs(B64).T                      # sum_k |a_k b_k|: what relative errors of the terms multiply


def acc32(pairs):
    out = np.zeros((M, N), np.float32)
    for k0 in range(0, D, 8):                          # fp32 accumulation, k in blocks (order does not matter for the comparison)
        for a, b in pairs:
            out = out + (a[:, k0:k0 + 8] @ b[:, k0:k0 + 8].T).astype(np.float32)
    return out.astype(np.float64)


fp32 = acc32([(A32, B32)])
a, b = split3(A32), split3(B32)
six = acc32([(a[0], b[0]), (a[0], b[1]), (a[1], b[0]), (a[0], b[2]), (a[1], b[1]), (a[2], b[0])])
three = acc32([(a[0], b[0]), (a[0], b[1]), (a[1], b[0])])
for name, got in (("fp32 operands, fp32 accumulation (product kernel)", fp32), ("bf16 x 3 terms, six products", six),
                  ("bf16 x 2 terms, three products", three)):
    e = np.abs(got - ref)
    print("%-52s max |err| %.2e  rms %.2e  max err / sum|a b| %.2e" % (name, e.max(), np.sqrt((e ** 2).mean()), (e / mag).max()))
print("scores of this size: |ref| up to %.1f; tolerance of the parity tests 1e-4 (+ 1e-4 relative)" % np.abs(ref).max())
