"""Dev: the opt-in bf16 three-way split products of the Newton root (PS_NEWTON_PRODUCTS=bf16x6)
against the exact-float32 products: step time, iteration counts, error vs the oracle / float64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from oracle import shampoo_oracle as orc
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
f = lambda x, y: float(np.linalg.norm(x - y) / np.linalg.norm(y))
for name in ("cfg2_256x512_p4", "headline_64x1024_p4"):
  st, p = bench.make_blocks(name, 0, dev)
  mats = list(st.unbind(0))
  res = {}
  for mode in ("f32", "bf16x6"):
    os.environ["PS_NEWTON_PRODUCTS"] = mode
    for _ in range(2):
      r, m = K.matrix_inverse_pth_root_batched(mats, [p] * len(mats))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
      r, m = K.matrix_inverse_pth_root_batched(mats, [p] * len(mats))
    torch.cuda.synchronize()
    res[mode] = ((time.perf_counter() - t0) / 5 * 1e3, [x.cpu().numpy() for x in r[:4]], m.cpu().numpy())
  a = mats[0].cpu().numpy()
  h_ref, m_ref = orc.matrix_inverse_pth_root(a, p)
  print(name, "f32 %.2f ms  bf16x6 %.2f ms | iters f32 %s x6 %s (oracle %d) | rel vs oracle f32 %.2e x6 %.2e | x6 vs f32 %.2e | max err metric %.2e / %.2e" % (
      res["f32"][0], res["bf16x6"][0], sorted(set(res["f32"][2][:, 1])), sorted(set(res["bf16x6"][2][:, 1])),
      m_ref["inverse_pth_root_iters"], f(res["f32"][1][0], h_ref), f(res["bf16x6"][1][0], h_ref),
      f(res["bf16x6"][1][0], res["f32"][1][0]), res["f32"][2][:, 0].max(), res["bf16x6"][2][:, 0].max()), flush=True)
  del st, mats
  torch.cuda.empty_cache()
# ill-conditioned sample: ViT-B blocks
vw = bench.VitBWorkload(0, 1, dev, None)
vw.stats_step()
flat = [s for st_ in vw.stats for s in st_]
seen = {}
for i, (s, p) in enumerate(zip(flat, vw.exps)):
  key = (int(s.shape[0]), p)
  if key in seen or key[0] < 700:
    continue
  seen[key] = 1
  a = s.cpu().numpy()
  w, v = np.linalg.eigh(a.astype(np.float64))
  out = []
  for mode in ("f32", "bf16x6"):
    os.environ["PS_NEWTON_PRODUCTS"] = mode
    r, m = K.matrix_inverse_pth_root_batched([s], [p])
    m = m.cpu().numpy()
    ridge = 1e-6 * float(m[0, 3])
    h64 = (v * (np.maximum(w, 0) + ridge) ** (-1.0 / p)) @ v.T
    out.append("%s: vs f64 %.2e iters %d err %.1e" % (mode, f(r[0].cpu().numpy(), h64), m[0, 1], m[0, 0]))
  print(key, " | ".join(out), flush=True)
