import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from oracle import shampoo_oracle as orc
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
vw = bench.VitBWorkload(0, 1, dev, None)
vw.stats_step()
flat = [s for st in vw.stats for s in st]
for i in (0, 23, 393, 28):
  a_d = flat[i]; p = vw.exps[i]
  a = a_d.cpu().numpy()
  w, v = np.linalg.eigh(a.astype(np.float64))
  f = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm(y)
  out = []
  for sym, navg in (("verify", "0"), ("verify", "2"), ("verify", "4"), ("general", "0")):
    os.environ["PS_NEWTON_AVG_STEPS"] = navg
    r, m = K.matrix_inverse_pth_root_batched([a_d], [p], symmetry=sym)
    m = m.cpu().numpy()
    ridge = 1e-6 * float(m[0, 3])
    h64 = (v * (np.maximum(w, 0) + ridge) ** (-1.0 / p)) @ v.T
    h = r[0].cpu().numpy()
    out.append("%s/avg%s: vs-f64 %.2e asym0 %.1e asym %.1e iters %d err %.1e" % (sym, navg, f(h, h64), m[0, 7], np.abs(h - h.T).max() / np.abs(h).max(), m[0, 1], m[0, 0]))
  h_ref, m_ref = orc.matrix_inverse_pth_root(a, p)
  out.append("oracle: vs-f64 %.2e asym %.1e" % (f(h_ref, h64), np.abs(h_ref - h_ref.T).max() / np.abs(h_ref).max()))
  hs = 0.5 * (h_ref + h_ref.T)
  out.append("oracle symmetrized: %.2e" % f(hs, h64))
  print(i, a.shape, p, " | ".join(out))
