"""Dev: accuracy of the eigh root on rank-deficient + ridge inputs vs float64, beside LAPACK
float32 (PS_EIGH_REFINE=0 switches the float64 Rayleigh-quotient refinement off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
for n, r in ((128, 30), (200, 50), (65, 16), (260, 260), (512, 100)):
  g = rng.standard_normal((n, r)); a = (g @ g.T).astype(np.float32)
  p = 4
  roots, m = K.matrix_inverse_pth_root_batched([torch.tensor(a, device=dev)], [p], [n], eigh=True)
  got = roots[0].cpu().numpy().astype(np.float64)
  a64 = a.astype(np.float64)
  lam = np.linalg.eigvalsh(a64).max(); ridge = 1e-6 * lam
  w, v = np.linalg.eigh(a64 + ridge * np.eye(n))
  f = lambda e: np.where(e == 0, 0.0, np.maximum(e, ridge) ** (-1.0 / p))
  truth = (v * f(w)) @ v.T
  d32 = (a + np.float32(ridge) * np.eye(n, dtype=np.float32)).astype(np.float32)
  wl, vl = np.linalg.eigh(d32)
  lap = (vl.astype(np.float64) * f(wl.astype(np.float64))) @ vl.T.astype(np.float64)
  tn = np.linalg.norm(truth)
  print(f"n={n} rank={r}: hip {np.linalg.norm(got - truth)/tn:.2e}  lapack f32 {np.linalg.norm(lap - truth)/tn:.2e}  err metric {float(m[0,0]):.2e}")
