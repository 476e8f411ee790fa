import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import shampoo_oracle as orc
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
def wishart(n, k, seed):
  g = np.random.default_rng(seed).standard_normal((n, k)).astype(np.float32)
  return (g @ g.T).astype(np.float32)
def run(md, ps, mode):
  os.environ["PS_NEWTON_PERSISTENT"] = mode
  r, m = K.matrix_inverse_pth_root_batched(md, ps)
  torch.cuda.synchronize()
  return [x.cpu().numpy() for x in r], m.cpu().numpy()
for p in (1, 2, 3, 4, 6, 8):
  for n in (33, 100, 200, 300):
    a = wishart(n, 4 * n, n + p)
    md = [torch.tensor(a, device=dev)]
    p1 = run(md, [p], "1"); p2 = run(md, [p], "1"); s1 = run(md, [p], "0")
    h64 = None
    w, v = np.linalg.eigh(a.astype(np.float64))
    lam = p1[1][0, 3]
    h64 = (v * (w + 1e-6 * lam) ** (-1.0 / p)) @ v.T
    e = lambda h: np.linalg.norm(h - h64) / np.linalg.norm(h64)
    print(f"p={p} n={n}: persist==persist {np.array_equal(p1[0][0], p2[0][0])} persist==staged "
          f"{np.array_equal(p1[0][0], s1[0][0])} metrics== {np.array_equal(p1[1], s1[1], equal_nan=True)} "
          f"err64 persist {e(p1[0][0]):.2e} staged {e(s1[0][0]):.2e} iters {p1[1][0,1]} {s1[1][0,1]}")
