"""Dev: error of the p=4 roots of the ViT-B sample blocks against the float64 closed form,
and the cfg2 / headline step time, as a function of PS_NEWTON_AVG_STEPS."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from oracle import shampoo_oracle as orc
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
vw = bench.VitBWorkload(0, 1, dev, None)
vw.stats_step()
flat = [s for st in vw.stats for s in st]
f = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm(y)
sample = {}
for i, (s, p) in enumerate(zip(flat, vw.exps)):
  sample.setdefault((int(s.shape[0]), p), []).append(i)
for key in ((1024, 4), (768, 4), (1000, 4), (768, 2), (1024, 2)):
  for i in sample[key][:2]:
    a_d = flat[i]; p = vw.exps[i]
    a = a_d.cpu().numpy()
    w, v = np.linalg.eigh(a.astype(np.float64))
    out = []
    for sym, navg in (("verify", "0"), ("verify", "2"), ("verify", "4"), ("verify", "6"), ("verify", "100"), ("general", "0")):
      os.environ["PS_NEWTON_AVG_STEPS"] = navg
      r, m = K.matrix_inverse_pth_root_batched([a_d], [p], symmetry=sym)
      m = m.cpu().numpy()
      ridge = 1e-6 * float(m[0, 3])
      h64 = (v * (np.maximum(w, 0) + ridge) ** (-1.0 / p)) @ v.T
      h = r[0].cpu().numpy()
      out.append("%s/avg%s %.2e it%d" % (sym[0], navg, f(h, h64), m[0, 1]))
    h_ref, m_ref = orc.matrix_inverse_pth_root(a, p)
    out.append("oracle %.2e it%d; cond %.1e" % (f(h_ref, h64), m_ref["inverse_pth_root_iters"], (w[-1] + ridge) / (max(w[0], 0) + ridge)))
    print(key, i, " | ".join(out), flush=True)

for name in ("cfg2_256x512_p4", "headline_64x1024_p4"):
  st, _ = bench.make_blocks(name, 0, dev)
  mats = list(st.unbind(0))
  for navg in ("0", "2", "4", "6", "100"):
    os.environ["PS_NEWTON_AVG_STEPS"] = navg
    for _ in range(2):
      K.matrix_inverse_pth_root_batched(mats, [4] * len(mats))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
      K.matrix_inverse_pth_root_batched(mats, [4] * len(mats))
    torch.cuda.synchronize()
    print(name, "avg", navg, "%.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
