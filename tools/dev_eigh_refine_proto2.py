"""Dev prototype 2: re-project A' = V^T D V with float64 accumulation, then float32 Jacobi
finishing sweeps on A', float64 Rayleigh quotients: how accurate does the root get?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
F = np.float32

def make(n, kind):
  if kind == "lowrank":
    g = rng.standard_normal((n, max(n // 4, 1))); a = g @ g.T
  else:
    q, _ = np.linalg.qr(rng.standard_normal((n, n))); e = 10.0 ** rng.uniform(-4, 2, n)
    a = (q * e) @ q.T
  return ((a + a.T) / 2).astype(F)

def jacobi_sweeps_f32(a, sweeps):
  """cyclic two-sided Jacobi in float32 (vectorised per rotation); returns (a, q)."""
  n = a.shape[0]; a = a.astype(F).copy(); q = np.eye(n, dtype=F)
  for _ in range(sweeps):
    for i in range(n - 1):
      for j in range(i + 1, n):
        apq = a[i, j]
        if apq == 0: continue
        tau = (a[j, j] - a[i, i]) / (F(2) * apq)
        t = np.sign(tau) / (abs(tau) + np.sqrt(F(1) + tau * tau)) if tau != 0 else F(1)
        c = F(1) / np.sqrt(F(1) + t * t); s = c * t
        ri, rj = a[i].copy(), a[j].copy()
        a[i], a[j] = c * ri - s * rj, s * ri + c * rj
        ci, cj = a[:, i].copy(), a[:, j].copy()
        a[:, i], a[:, j] = c * ci - s * cj, s * ci + c * cj
        qi, qj = q[:, i].copy(), q[:, j].copy()
        q[:, i], q[:, j] = c * qi - s * qj, s * qi + c * qj
  return a, q

p = 4
for n, kind in ((129, "graded"), (200, "lowrank")):
  a = make(n, kind); a64 = a.astype(np.float64)
  ridge = 1e-6 * np.linalg.eigvalsh(a64).max()
  d64 = a64 + ridge * np.eye(n); d32 = d64.astype(F)
  w, v = np.linalg.eigh(d64)
  f = lambda e: np.where(e == 0, 0.0, np.maximum(e, ridge) ** (-1.0 / p))
  truth = (v * f(w)) @ v.T; tn = np.linalg.norm(truth)
  wl, vl = np.linalg.eigh(d32)
  lap = (vl.astype(np.float64) * f(wl.astype(np.float64))) @ vl.T.astype(np.float64)
  es, vs = K.eigh_batched([torch.tensor(d32, device=dev)])
  x = vs[0].cpu().numpy()
  def root_from(xv):
    xv64 = xv.astype(np.float64)
    lam = np.einsum("ij,ij->j", xv64, d32.astype(np.float64) @ xv64) / np.einsum("ij,ij->j", xv64, xv64)
    return (xv64 * f(lam)) @ xv64.T
  line = [f"n={n} {kind}: lapack {np.linalg.norm(lap-truth)/tn:.1e} hip+rq64 {np.linalg.norm(root_from(x)-truth)/tn:.1e}"]
  for acc in ("f32", "f64"):
    if acc == "f32":
      ap = (x.T @ (d32 @ x)).astype(F)
    else:
      ap = (x.astype(np.float64).T @ (d32.astype(np.float64) @ x.astype(np.float64))).astype(F)
    ap = ((ap + ap.T) / 2).astype(F)
    for sw in (2, 4, 6):
      _, q = jacobi_sweeps_f32(ap, sw)
      x2 = (x.astype(np.float64) @ q.astype(np.float64)).astype(F)
      line.append(f"reproj {acc}+{sw}sw {np.linalg.norm(root_from(x2)-truth)/tn:.1e}")
  print("  ".join(line))
