"""dev: LDS-resident eigensolver (n <= 128) vs float64 LAPACK; timing of batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
def make(n, kind):
  if kind == "wishart":
    g = rng.standard_normal((n, 3 * n)); return (g @ g.T).astype(np.float32)
  if kind == "indef":
    a = rng.standard_normal((n, n)); return ((a + a.T) / 2).astype(np.float32)
  if kind == "graded":
    q, _ = np.linalg.qr(rng.standard_normal((n, n))); e = 10.0 ** (-6 * np.arange(n) / max(n - 1, 1))
    a = (q * e) @ q.T; return ((a + a.T) / 2).astype(np.float32)
  if kind == "lowrank":
    g = rng.standard_normal((n, max(n // 4, 1))); return (g @ g.T).astype(np.float32)
for n in (1, 2, 5, 33, 64, 96, 127, 128):
  for kind in ("wishart", "indef", "graded", "lowrank"):
    a = make(n, kind)
    es, vs = K.eigh_batched([torch.tensor(a, device=dev)])
    e, v = es[0].cpu().numpy(), vs[0].cpu().numpy()
    w = np.linalg.eigvalsh(a.astype(np.float64))
    scale = np.abs(w).max() + 1e-30
    print(f"n={n:3d} {kind:8s} eval err {np.abs(e - w).max() / scale:.1e} orth {np.abs(v.T @ v - np.eye(n)).max():.1e} "
          f"resid {np.abs(a @ v - v * e).max() / scale:.1e}")
for batch, n in ((8, 96), (8, 128), (64, 96), (256, 64), (395, 128)):
  mats = [torch.tensor(make(n, "wishart"), device=dev) for _ in range(batch)]
  for mode in ("1", "0"):
    os.environ["PS_EIGH_SMALL"] = mode
    K.eigh_batched(mats); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
      K.eigh_batched(mats)
    torch.cuda.synchronize()
    print(f"batch {batch} x {n}: small={mode} {(time.perf_counter() - t0) / 3 * 1e3:.3f} ms per call")
