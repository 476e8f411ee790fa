"""Dev: accuracy of the eigh root on the cases of test_eigh_root_accuracy_on_graded_spectra_near_lapack
(and larger ones) for combinations of the Cholesky-Jacobi switches."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
cases = [(169, "graded", 4), (512, "graded", 2), (260, "lowrank", 2), (1024, "graded", 4),
         (2048, "graded", 2), (2048, "lowrank", 2), (1000, "wishart", 2)]
for cfg in sys.argv[1:] or ["default"]:
  env = dict(kv.split("=") for kv in cfg.split(",") if "=" in kv)
  os.environ.update(env)
  out = []
  for n, kind, p in cases:
    rng = np.random.default_rng(n + p)
    if kind == "lowrank":
      g = rng.standard_normal((n, n // 4)); a = g @ g.T
    elif kind == "wishart":
      g = rng.standard_normal((n, 2 * n)); a = g @ g.T
    else:
      q, _ = np.linalg.qr(rng.standard_normal((n, n)))
      a = (q * 10.0 ** rng.uniform(-4, 2, n)) @ q.T
    a = ((a + a.T) / 2).astype(np.float32)
    a64 = a.astype(np.float64)
    ridge = 1e-6 * np.linalg.eigvalsh(a64).max()
    w, v = np.linalg.eigh(a64 + ridge * np.eye(n))
    f = lambda e: np.maximum(e, ridge) ** (-1.0 / p)
    truth = (v * f(w)) @ v.T
    d32 = (a + np.float32(ridge) * np.eye(n, dtype=np.float32)).astype(np.float32)
    wl, vl = np.linalg.eigh(d32)
    lap = (vl.astype(np.float64) * f(wl.astype(np.float64))) @ vl.T.astype(np.float64)
    roots, m = K.matrix_inverse_pth_root_batched([torch.tensor(a, device=dev)], [p], [n], eigh=True)
    got = roots[0].cpu().numpy().astype(np.float64)
    tn = np.linalg.norm(truth)
    out.append("%d/%s/p%d: %.2e (lap %.2e) sw%d" % (n, kind[:2], p, np.linalg.norm(got - truth) / tn,
                                                  np.linalg.norm(lap - truth) / tn, m[0, 5].item()))
  print(cfg, " | ".join(out), flush=True)
  for k in env:
    os.environ.pop(k)
