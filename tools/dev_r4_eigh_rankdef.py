"""dev (GPU): eigh root on rank-deficient + ridge inputs (statistics of a few gradient outer products):
error vs the float64 closed form for the one-sided (default) and the two-sided solver and LAPACK float32."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import shampoo_oracle as orc
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
rng = np.random.default_rng(11)
for n in (200, 260, 512, 1024):
  for frac in (4, 2):
    for p in (2, 4):
      g = rng.standard_normal((n, n // frac)); a = (g @ g.T).astype(np.float32); a = (a + a.T) / 2
      a64 = a.astype(np.float64); w = np.linalg.eigvalsh(a64); mx = w.max()
      ridge = 1e-6 * mx
      w2, v2 = np.linalg.eigh(a64 + ridge * np.eye(n))
      truth = (v2 * np.maximum(w2, ridge) ** (-1.0 / p)) @ v2.T; tn = np.linalg.norm(truth)
      h, _ = orc.matrix_inverse_pth_root_eigh(a, p, padding_start=n)
      out = []
      for env in ({}, {"PS_EIGH_CJ": "0"}):
        os.environ.pop("PS_EIGH_CJ", None); os.environ.update(env)
        r, met = K.matrix_inverse_pth_root_batched([torch.tensor(a, device=dev)], [p], [n], eigh=True)
        out.append(np.linalg.norm(r[0].cpu().numpy() - truth) / tn)
      os.environ.pop("PS_EIGH_CJ", None)
      e_l = np.linalg.norm(h - truth) / tn
      print(f"n {n:5d} rank n/{frac} p {p}: lapack32 {e_l:.2e}  one-sided {out[0]:.2e} ({out[0] / e_l:4.1f}x)  two-sided {out[1]:.2e} ({out[1] / e_l:4.1f}x)", flush=True)
