"""Dev (round 5): plain eigenpairs (ps_eigh_batched_opt_f32, default solver) on mixed batches: blocks the fast path keeps,
blocks it hands to the two-sided Jacobi path and blocks of the small solver in ONE call, against NumPy float64."""
import os
os.environ.setdefault("PS_DEV_ENV", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for batch in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
  mats, kinds = [], []
  for _ in range(int(rng.integers(2, 9))):
    n = int(rng.choice([1, 5, 64, 128, 129, 150, 193, 200, 300, 520]))
    kind = str(rng.choice(["wishart", "indef", "graded", "lowrank", "wishart"]))
    if kind == "wishart":
      g = rng.standard_normal((n, 2 * n + 1)); a = g @ g.T
    elif kind == "indef":
      g = rng.standard_normal((n, n)); a = g + g.T
    elif kind == "lowrank":
      g = rng.standard_normal((n, max(1, n // 4))); a = g @ g.T
    else:
      q, _ = np.linalg.qr(rng.standard_normal((n, n))); a = (q * 10.0 ** rng.uniform(-4, 1, n)) @ q.T
    mats.append(((a + a.T) / 2).astype(np.float32)); kinds.append(kind)
  es, vs = K.eigh_batched([torch.tensor(a, device=dev) for a in mats])
  for i, (a, kind, e, v) in enumerate(zip(mats, kinds, es, vs)):
    a64 = a.astype(np.float64); w = np.linalg.eigvalsh(a64); nrm = max(np.abs(w).max(), 1e-300); n = a.shape[0]
    e = e.cpu().numpy().astype(np.float64); v = v.cpu().numpy().astype(np.float64)
    ev = np.abs(e - w).max() / nrm; res = np.abs(a64 @ v - v * e).max() / nrm; orth = np.abs(v.T @ v - np.eye(n)).max()
    # small eigenvalues of graded / rank-deficient inputs: relative accuracy where they are above eps * norm
    ok = ev <= 3e-6 and res <= 6e-6 and orth <= 2e-5 and np.isfinite(v).all()
    if not ok:
      bad += 1
      print(f"batch {batch} block {i}: n={n} {kind}: ev={ev:.2e} res={res:.2e} orth={orth:.2e}   <-- CHECK", flush=True)
print("plain eigh fuzz mismatches", bad)
