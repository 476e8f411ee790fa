import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from oracle import shampoo_oracle as orc
from precondition_amd import comm
dev = torch.device("cuda:0")
vw = bench.VitBWorkload(0, 1, dev, None)
vw.stats_step()
flat = [s for st in vw.stats for s in st]
roots, metrics = comm.sharded_inverse_pth_roots(flat, vw.exps, group=None, ownership="lpt", pi_first=True)
met = metrics.cpu().numpy()
sample = {}
for i, (s, p) in enumerate(zip(flat, vw.exps)):
  sample.setdefault((int(s.shape[0]), p), i)
for key, i in sample.items():
  a = flat[i].cpu().numpy(); p = key[1]
  h_ref, m_ref = orc.matrix_inverse_pth_root(a, p)
  h = roots[i].cpu().numpy()
  w, v = np.linalg.eigh(a.astype(np.float64))
  ridge = 1e-6 * float(met[i, 3])
  h64 = (v * (np.maximum(w, 0) + ridge) ** (-1.0 / p)) @ v.T
  f = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm(y)
  print(key, "idx", i, "cond", (w.max() + ridge) / (max(w.min(), 0) + ridge), "hip-vs-oracle %.2e hip-vs-f64 %.2e oracle-vs-f64 %.2e" % (f(h, h_ref), f(h, h64), f(h_ref, h64)),
        "iters", met[i, 1], m_ref["inverse_pth_root_iters"], "retries", met[i, 4], m_ref["total_retries"], "err", met[i, 0], m_ref["inverse_pth_root_errors"])
