import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
def make(n):
  g = rng.standard_normal((n, 3 * n)); return (g @ g.T).astype(np.float32)
for batch, n in ((8, 96), (8, 128), (256, 64)):
  mats = [torch.tensor(make(n), device=dev) for _ in range(batch)]
  for _ in range(2):
    K.eigh_batched(mats)
  torch.cuda.synchronize()
