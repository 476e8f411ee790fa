import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from precondition_amd import kernels as K, low_rank
dev = torch.device("cuda:0")
np.set_printoptions(precision=4, linewidth=200)
rng = np.random.default_rng(0)
for n, k in ((12, 1), (32, 1), (32, 3), (100, 2)):
  g = rng.standard_normal((n, k)).astype(np.float32)
  gram = torch.tensor(g @ g.T, device=dev)
  (e,), (u,) = K.eigh_batched([gram])
  e = e.cpu().numpy()
  print(n, k, "lam_max", e.max(), "noise eigs (abs max of rest)/lam_max", np.abs(e[:-k]).max() / e.max(), "n*1.2e-7", n * 1.2e-7)
