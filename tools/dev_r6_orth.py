"""Dev (round 6): where does the eigenvectors' orthogonality go on the tridiagonalisation path?
orth(Q) after the reduction + back-transformation alone (PS_EIGH_TD_STAGE=1: Z_T = I), orth(Z) of the full solver,
and the same for a true float32 ssyevd (scipy) -- the error metric max|U^T D U - diag(e)| (DS:1017-1021) is
orth x lambda_max."""
import os
os.environ.setdefault("PS_DEV_ENV", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import lapack32
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
for n in (1024, 2048):
  g = np.random.default_rng(n).standard_normal((n, 2 * n)).astype(np.float32)
  a = (g @ g.T).astype(np.float32)
  t = torch.tensor(a, device=dev)
  line = f"n={n}"
  for stage in ("1", "0"):
    os.environ["PS_EIGH_TD_STAGE"] = stage
    e, v = K.eigh_batched([t], options={"eigh_solver": "tridiagonal"})
    q = v[0].double()
    orth = float((q.T @ q - torch.eye(n, device=dev, dtype=torch.float64)).abs().max())
    fro = float((q.T @ q - torch.eye(n, device=dev, dtype=torch.float64)).norm())
    line += f" | {'Q (reduction + WY)' if stage == '1' else 'Z (full)'}: max {orth:.2e} fro {fro:.2e}"
  os.environ.pop("PS_EIGH_TD_STAGE")
  w, v = lapack32.eigh32(a)
  v = v.astype(np.float64)
  d = v.T @ v - np.eye(n)
  print(line + f" | ssyevd: max {np.abs(d).max():.2e} fro {np.linalg.norm(d):.2e}", flush=True)
