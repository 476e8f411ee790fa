"""Dev: per-workgroup timeline of the Newton product kernel (PS_NEWTON_TRACE): phase lengths of
every tile (prologue = start -> first LDS fill, K loop, epilogue) and how the two workgroups of a
CU overlap."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_256x512_p4"
path = "/tmp/stage_trace.bin"
st, p = bench.make_blocks(name, 0, dev)
mats = list(st.unbind(0))
for _ in range(3):
  r, m = K.matrix_inverse_pth_root_batched(mats, [p] * len(mats))
torch.cuda.synchronize()
if os.path.exists(path): os.remove(path)
os.environ["PS_NEWTON_TRACE"] = path
r, m = K.matrix_inverse_pth_root_batched(mats, [p] * len(mats))
torch.cuda.synchronize()
del os.environ["PS_NEWTON_TRACE"]
rec = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
print(name, "records", len(rec))
seq = rec[:, 0].astype(np.int64); prod = (rec[:, 1] >> np.uint64(32)).astype(np.int64)
hw = (rec[:, 2] & np.uint64(0xffffffff)).astype(np.int64); xcc = (rec[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
t0, tf, tk, t3 = (rec[:, c].astype(np.int64) for c in (3, 4, 5, 6))
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
us = lambda x: x / 100.0
for s in sorted(set(seq.tolist()))[:12]:
  k = (seq == s) & (tk > 0)
  if k.sum() == 0:
    print("launch", s, "no product tiles"); continue
  base = t0[seq == s].min()
  span = us(t3[seq == s].max() - base)
  pro, kl, ep = us(tf[k] - t0[k]), us(tk[k] - tf[k]), us(t3[k] - tk[k])
  print("launch %2d prod %s tiles %5d span %7.1f us | prologue %5.1f (p90 %5.1f)  K loop %5.1f (p10 %5.1f p90 %5.1f)  epilogue %5.1f (p90 %5.1f) | CUs %d" % (
      s, sorted(set(prod[k].tolist())), k.sum(), span, pro.mean(), np.percentile(pro, 90), kl.mean(),
      np.percentile(kl, 10), np.percentile(kl, 90), ep.mean(), np.percentile(ep, 90), len(set(cuid[k].tolist()))))
# one launch in detail: fraction of the span in which a CU has 0 / 1 / 2 workgroups inside their K loops
s = int(sys.argv[2]) if len(sys.argv) > 2 else sorted(set(seq.tolist()))[4]
k = (seq == s) & (tk > 0)
base = t0[k].min(); end = t3[k].max()
grid = np.arange(base, end)  # 10 ns ticks
tot = np.zeros(3)
for c in sorted(set(cuid[k].tolist())):
  kk = k & (cuid == c)
  n_in = np.zeros(len(grid), dtype=np.int32)
  for a, b in zip(tf[kk], tk[kk]):
    n_in[a - base:b - base] += 1
  for j in range(3): tot[j] += (n_in == j).sum() if j < 2 else (n_in >= 2).sum()
tot /= tot.sum()
print("launch", s, ": share of CU-time with 0 / 1 / >=2 workgroups inside a K loop: %.3f %.3f %.3f" % tuple(tot))
# start-time histogram of the tiles of that launch (lockstep?)
st_us = us(t0[k] - base); en_us = us(t3[k] - base)
h, edges = np.histogram(st_us, bins=20)
print("tile start times (us) histogram:", list(zip(edges[:-1].round(0).tolist(), h.tolist())))
h, edges = np.histogram(us(tk[k] - base), bins=20)
print("K-loop end times (us) histogram:", list(zip(edges[:-1].round(0).tolist(), h.tolist())))
