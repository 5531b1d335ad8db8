"""Accuracy of emulating the f32 GEMMs of the PPO gradient pass (K7) with bf16 matrix instructions.
a = a_hi + a_lo (+ a_lo2) with bf16 pieces, products accumulated in float32:
  x3: a_hi b_hi + a_hi b_lo + a_lo b_hi                 (3 bf16 MFMAs per f32 MFMA-equivalent, ~16-bit operands)
  x6: + a_hi b_lo2 + a_lo2 b_hi + a_lo b_lo             (6 MFMAs, ~24-bit operands)
Compared with plain float32 accumulation against a float64 reference, on the shapes of the MLP (64x64 layer, contraction
over 64 for activations and over the samples for weight gradients)."""
import numpy as np

rng = np.random.default_rng(0)


def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF                      # round to nearest even
    return ((u + r) & 0xFFFF0000).view(np.float32)


def split(x, n):
    parts, rem = [], x.astype(np.float32)
    for _ in range(n):
        p = bf16(rem)
        parts.append(p)
        rem = (rem - p).astype(np.float32)
    return parts


def mm32(a, b):
    return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float32)


def emul(a, b, terms):
    A, B = split(a, 3), split(b, 3)
    acc = np.zeros((a.shape[0], b.shape[1]), dtype=np.float32)
    for i, j in terms:
        acc = (acc + mm32(A[i], B[j])).astype(np.float32)
    return acc


X3 = [(0, 0), (0, 1), (1, 0)]
X6 = X3 + [(0, 2), (2, 0), (1, 1)]
for name, M, K, N, scale in (("forward 64x64 layer", 64, 64, 4096, 1.0), ("dW2 = dpre2 . h1^T over 32768 samples", 64, 32768, 64, 1.0)):
    a = (rng.standard_normal((M, K)) * (0.2 if K == 64 else 1e-3)).astype(np.float32)
    b = np.tanh(rng.standard_normal((K, N))).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    s = np.abs(ref).max()
    for label, got in (("float32", mm32(a, b)), ("bf16 x3", emul(a, b, X3)), ("bf16 x6", emul(a, b, X6)), ("bf16 x1", mm32(bf16(a), bf16(b)))):
        err = np.abs(got - ref)
        print(f"{name:42s} {label:8s} max|err|/max|ref| = {err.max() / s:.2e}   rms rel = {np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean()):.2e}")
