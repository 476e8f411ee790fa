"""Dev: accuracy of the eigh inverse root on graded / rank-deficient inputs vs float64, beside
LAPACK float32, for sizes on the blocked path."""
import os, sys, time
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
for n, kind in ((64, "graded"), (96, "graded"), (128, "graded"), (128, "lowrank"), (100, "lowrank"), (129, "graded"), (169, "graded"), (260, "lowrank"), (512, "lowrank"), (512, "graded"),
                (1024, "graded"), (1024, "lowrank"), (512, "wishart"), (2048, "graded")):
  for p in (2, 4):
    a = make(n, kind); a64 = a.astype(np.float64)
    t0 = time.perf_counter()
    roots, m = K.matrix_inverse_pth_root_batched([torch.tensor(a, device=dev)], [p], [n], eigh=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    got = roots[0].cpu().numpy().astype(np.float64)
    lam = np.linalg.eigvalsh(a64).max(); ridge = 1e-6 * lam
    w, v = np.linalg.eigh(a64 + ridge * np.eye(n))
    f = lambda e: np.where(e == 0, 0.0, np.maximum(e, ridge) ** (-1.0 / p))
    truth = (v * f(w)) @ v.T
    d32 = (a + np.float32(ridge) * np.eye(n, dtype=np.float32)).astype(np.float32)
    wl, vl = np.linalg.eigh(d32)
    lap = (vl.astype(np.float64) * f(wl.astype(np.float64))) @ vl.T.astype(np.float64)
    tn = np.linalg.norm(truth)
    print(f"n={n} {kind} p={p}: hip {np.linalg.norm(got - truth)/tn:.2e}  lapack f32 {np.linalg.norm(lap - truth)/tn:.2e}  sweeps {float(m[0,5]):.0f}  {dt*1e3:.0f} ms")
