"""Dev (round 6): which blocks does the tridiagonalisation path hand back for an iteration cap, and what does
the deflation tolerance of the divide and conquer do to that and to the root error?

Statistics of rank-8 gradients after a few updates from matrix_epsilon * I (the state of a ViT-B tree's first
recomputes): rank 8 t + an EXACT multiple of the identity on the complement -- one huge cluster."""
import os
os.environ.setdefault("PS_DEV_ENV", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import shampoo_oracle as orc
from precondition_amd import kernels as K

dev = torch.device("cuda:0")


def stat(n, updates, seed, k=8):
  rng = np.random.default_rng(seed)
  a = np.float32(1e-6) * np.eye(n, dtype=np.float32)
  for _ in range(updates):
    g = (rng.standard_normal((n, k)) @ rng.standard_normal((k, 3 * n)) * 0.02 / 3).astype(np.float32)
    a = (np.float32(0.999) * a + np.float32(0.001) * (g @ g.T)).astype(np.float32)
  return ((a + a.T) / 2).astype(np.float32)


for n in (768, 1024):
  for updates in (1, 3, 7, 13):
    a = stat(n, updates, n + updates)
    p = 4
    truth = orc.eigh_root_float64(a, p)
    tn = np.linalg.norm(truth)
    h, _ = orc.matrix_inverse_pth_root_eigh(a, p)
    line = f"n={n} updates={updates:2d} ssyevd {np.linalg.norm(h - truth) / tn:.2e} |"
    for eps in ("1e-8", "3e-8", "6e-8"):
      os.environ["PS_EIGH_TD_DEFL_EPS"] = eps
      r, m = K.matrix_inverse_pth_root_batched([torch.tensor(a, device=dev)], [p], eigh=True,
                                               options={"eigh_solver": "tridiagonal"})
      m = m.cpu().numpy()
      e = np.linalg.norm(r[0].cpu().numpy().astype(np.float64) - truth) / tn
      line += f" defl {eps}: e={e:.2e} sweeps={m[0, 5]:.0f} |"
    os.environ.pop("PS_EIGH_TD_DEFL_EPS")
    r, m = K.matrix_inverse_pth_root_batched([torch.tensor(a, device=dev)], [p], eigh=True,
                                             options={"eigh_solver": "one_sided"})
    e = np.linalg.norm(r[0].cpu().numpy().astype(np.float64) - truth) / tn
    print(line + f" one_sided e={e:.2e}", flush=True)
