import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
def wishart(n, k, seed):
  g = np.random.default_rng(seed).standard_normal((n, k)).astype(np.float32)
  return (g @ g.T).astype(np.float32)
def run(md, ps, mode, it):
  os.environ["PS_NEWTON_PERSISTENT"] = mode
  r, m = K.matrix_inverse_pth_root_batched(md, ps, num_iters=it)
  torch.cuda.synchronize()
  return [x.cpu().numpy() for x in r], m.cpu().numpy()
for p, n in ((6, 33), (3, 100)):
  a = wishart(n, 4 * n, n + p)
  md = [torch.tensor(a, device=dev)]
  for it in range(1, 8):
    p1 = run(md, [p], "1", it); s1 = run(md, [p], "0", it)
    d = np.abs(p1[0][0] - s1[0][0])
    print(f"p={p} n={n} num_iters={it}: equal {np.array_equal(p1[0][0], s1[0][0])} maxdiff {d.max():.2e} "
          f"ndiff {(d>0).sum()} of {d.size} tries {p1[1][0,4]} {s1[1][0,4]} it {p1[1][0,1]} err {p1[1][0,0]:.3e} {s1[1][0,0]:.3e}")
    if it == 1:
      i, j = np.unravel_index(np.argmax(d), d.shape)
      print("   at", i, j, p1[0][0][i, j], s1[0][0][i, j], " diag ratio", (p1[0][0].diagonal()/s1[0][0].diagonal())[:4])
