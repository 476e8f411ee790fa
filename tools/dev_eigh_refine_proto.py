"""Dev prototype: does one Ogita-Aishima refinement step in float64 on the blocked-Jacobi
eigenvectors bring the inverse root to (better than) LAPACK-float32 accuracy?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)

def make(n, kind):
  if kind == "lowrank":
    g = rng.standard_normal((n, max(n // 4, 1))); a = g @ g.T
  elif kind == "graded":
    q, _ = np.linalg.qr(rng.standard_normal((n, n))); e = 10.0 ** rng.uniform(-4, 2, n)
    a = (q * e) @ q.T
  else:
    g = rng.standard_normal((n, 2 * n)); a = g @ g.T
  return ((a + a.T) / 2).astype(np.float32)

def refine(a64, x, steps=1):
  """Ogita & Aishima: X' = X + X E, E from R = I - X^T X and S = X^T A X."""
  n = a64.shape[0]
  for _ in range(steps):
    r = np.eye(n) - x.T @ x
    s = x.T @ (a64 @ x)
    lam = np.diag(s) / (1.0 - np.diag(r))
    d = lam[None, :] - lam[:, None]          # lam_j - lam_i
    num = s + r * lam[None, :]
    delta = 2.0 * (np.linalg.norm(s - np.diag(lam)) + np.linalg.norm(a64, 2) * np.linalg.norm(r))
    with np.errstate(all="ignore"):
      e = np.where(np.abs(d) > delta, num / d, r / 2.0)
    e[np.arange(n), np.arange(n)] = np.diag(r) / 2.0
    x = x + x @ e
  s = x.T @ (a64 @ x)
  return np.diag(s) / np.diag(x.T @ x), x

p = 4
for n, kind in ((129, "graded"), (169, "graded"), (260, "lowrank"), (512, "lowrank"), (512, "graded"),
                (1024, "graded"), (1024, "lowrank"), (512, "wishart")):
  a = make(n, kind)
  a64 = a.astype(np.float64)
  lam_max = np.linalg.eigvalsh(a64).max(); ridge = 1e-6 * lam_max
  d64 = a64 + ridge * np.eye(n)
  w, v = np.linalg.eigh(d64)
  f = lambda e: np.where(e == 0, 0.0, np.maximum(e, ridge) ** (-1.0 / p))
  truth = (v * f(w)) @ v.T; tn = np.linalg.norm(truth)
  d32 = d64.astype(np.float32)
  wl, vl = np.linalg.eigh(d32)
  lap = (vl.astype(np.float64) * f(wl.astype(np.float64))) @ vl.T.astype(np.float64)
  es, vs = K.eigh_batched([torch.tensor(d32, device=dev)])
  e_h, x_h = es[0].cpu().numpy().astype(np.float64), vs[0].cpu().numpy().astype(np.float64)
  hip = (x_h * f(e_h)) @ x_h.T
  out = [f"n={n} {kind}: lapack {np.linalg.norm(lap - truth)/tn:.1e} hip {np.linalg.norm(hip - truth)/tn:.1e}"]
  for steps in (1, 2):
    e_r, x_r = refine(d32.astype(np.float64), x_h, steps)
    ref = (x_r * f(e_r)) @ x_r.T
    out.append(f"refine{steps} {np.linalg.norm(ref - truth)/tn:.1e}")
  # float32 storage of the refined vectors (what the kernel would hand on)
  e_r, x_r = refine(d32.astype(np.float64), x_h, 1)
  x32 = x_r.astype(np.float32).astype(np.float64)
  out.append(f"refine1->f32 vecs {np.linalg.norm((x32 * f(e_r)) @ x32.T - truth)/tn:.1e}")
  print("  ".join(out))
