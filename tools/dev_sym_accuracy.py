"""dev (CPU, numpy float32): how the treatment of the products' asymmetric rounding noise
affects the accuracy of the coupled Newton iteration at cond ~ 7e3, p = 4.
variants: full (reference), mirror (upper triangle copied to lower, what the symmetric
HIP mode did in round 1), mirror_squares_only, avg (mixed products fully computed then
(X + X^T)/2, squares mirrored), avg_all."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import numpy as np
F32 = np.float32
rng = np.random.default_rng(0)
n, m = 1000, 768
S = (1e-6 * 0.999 ** 5) * np.eye(n, dtype=F32)
G = (rng.standard_normal((m, n)) * 0.02).astype(F32)   # the same gradient every step (bench.py)
for t in range(5):
  S = (F32(0.999) * S + F32(0.001) * (G.T @ G)).astype(F32)
S = ((S + S.T) / 2).astype(F32)
w, v = np.linalg.eigh(S.astype(np.float64))
lam = w.max(); ridge = 1e-6 * lam
p = 4
h64 = (v * (w + ridge) ** (-1.0 / p)) @ v.T
print("cond", (w.max() + ridge) / (w.min() + ridge))

def mirror(x):
  u = np.triu(x)
  return (u + np.triu(x, 1).T).astype(F32)
def avg(x):
  return ((x + x.T) * F32(0.5)).astype(F32)

def run(mode):
  ident = np.eye(n, dtype=F32)
  alpha = F32(-1.0 / p); oma = F32(1) - alpha
  damped = (S + F32(ridge) * ident).astype(F32)
  z = F32(1 + p) / (F32(2) * np.linalg.norm(damped))
  M = (damped * z).astype(F32)
  H = (ident * np.power(z, F32(1.0 / p))).astype(F32)
  err = np.max(np.abs(M - ident)); it = 0
  sq = {"full": lambda x: x, "mirror": mirror, "mirror_squares_only": mirror, "avg": mirror, "avg_all": avg}[mode]
  mx = {"full": lambda x: x, "mirror": mirror, "mirror_squares_only": lambda x: x, "avg": avg, "avg_all": avg}[mode]
  while it < 100 and err > 1e-6:
    Mi = (oma * ident + alpha * M).astype(F32)
    T0 = sq(Mi @ Mi); T1 = sq(T0 @ T0)
    Mn = mx(T1 @ M); Hn = mx(H @ Mi)
    nerr = np.max(np.abs(Mn - ident))
    if nerr / err >= 1.2: break
    M, H, err, it = Mn, Hn, nerr, it + 1
  return H, it, err
for mode in ("full", "mirror", "mirror_squares_only", "avg", "avg_all"):
  H, it, err = run(mode)
  print(f"{mode:22s} iters {it} err {err:.2e} vs-f64 {np.linalg.norm(H - h64) / np.linalg.norm(h64):.3e} asym {np.abs(H - H.T).max() / np.abs(H).max():.1e}")

def mirror_lo(x):
  return mirror(x.T)
def run2(name, fH, fM):
  """fH(it), fM(it) -> symmetriser for the mixed products H*Mi and T1*M at step it."""
  ident = np.eye(n, dtype=F32)
  alpha = F32(-1.0 / p); oma = F32(1) - alpha
  damped = (S + F32(ridge) * ident).astype(F32)
  z = F32(1 + p) / (F32(2) * np.linalg.norm(damped))
  M = (damped * z).astype(F32)
  H = (ident * np.power(z, F32(1.0 / p))).astype(F32)
  err = np.max(np.abs(M - ident)); it = 0
  while it < 100 and err > 1e-6:
    Mi = (oma * ident + alpha * M).astype(F32)
    T0 = mirror(Mi @ Mi); T1 = mirror(T0 @ T0)
    Mn = fM(it)(T1 @ M); Hn = fH(it)(H @ Mi)
    nerr = np.max(np.abs(Mn - ident))
    if nerr / err >= 1.2: break
    M, H, err, it = Mn, Hn, nerr, it + 1
  print(f"{name:34s} iters {it} err {err:.2e} vs-f64 {np.linalg.norm(H - h64) / np.linalg.norm(h64):.3e}")
up, lo = mirror, mirror_lo
run2("H up, M up (round 1)", lambda it: up, lambda it: up)
run2("H alt, M alt (same phase)", lambda it: up if it % 2 == 0 else lo, lambda it: up if it % 2 == 0 else lo)
run2("H up, M lo", lambda it: up, lambda it: lo)
run2("H alt, M anti-alt", lambda it: up if it % 2 == 0 else lo, lambda it: lo if it % 2 == 0 else up)
run2("H avg, M up", lambda it: avg, lambda it: up)
run2("H up, M avg", lambda it: up, lambda it: avg)
run2("H avg, M avg", lambda it: avg, lambda it: avg)
# operand order: M' = M T1 instead of T1 M, H' = Mi H instead of H Mi (mirrored upper)
def run3(name, swapM, swapH):
  ident = np.eye(n, dtype=F32)
  alpha = F32(-1.0 / p); oma = F32(1) - alpha
  damped = (S + F32(ridge) * ident).astype(F32)
  z = F32(1 + p) / (F32(2) * np.linalg.norm(damped))
  M = (damped * z).astype(F32)
  H = (ident * np.power(z, F32(1.0 / p))).astype(F32)
  err = np.max(np.abs(M - ident)); it = 0
  while it < 100 and err > 1e-6:
    Mi = (oma * ident + alpha * M).astype(F32)
    T0 = mirror(Mi @ Mi); T1 = mirror(T0 @ T0)
    Mn = mirror(M @ T1 if swapM else T1 @ M); Hn = mirror(Mi @ H if swapH else H @ Mi)
    nerr = np.max(np.abs(Mn - ident))
    if nerr / err >= 1.2: break
    M, H, err, it = Mn, Hn, nerr, it + 1
  print(f"{name:34s} iters {it} err {err:.2e} vs-f64 {np.linalg.norm(H - h64) / np.linalg.norm(h64):.3e}")
run3("mirror, M T1 / H Mi", True, False)
run3("mirror, T1 M / Mi H", False, True)
run3("mirror, M T1 / Mi H", True, True)
print("---- M-update averaged only in some steps (H mirrored)")
for k0 in (0, 4, 8, 10, 12, 14):
  run2(f"M avg for it >= {k0}", lambda it: up, lambda it, k0=k0: avg if it >= k0 else up)
for k0 in (4, 8, 12):
  run2(f"M avg for it < {k0}", lambda it: up, lambda it, k0=k0: avg if it < k0 else up)
print("---- how many leading steps need the averaged M-update")
for k0 in (1, 2, 3):
  run2(f"M avg for it < {k0}", lambda it: up, lambda it, k0=k0: avg if it < k0 else up)
