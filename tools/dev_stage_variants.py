"""Dev: A/B of the Newton stage-kernel variants (PS_NEWTON_BK / PS_NEWTON_DEEP are read once per
process, so each variant runs in a child process)."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, subprocess, sys
CHILD = r'''
import sys; sys.path.insert(0, ".")
import torch, bench, time, os
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
for name in ("cfg2_256x512_p4", "headline_64x1024_p4"):
  st, p = bench.make_blocks(name, 0, dev); mats = list(st.unbind(0))
  for _ in range(3): r, m = K.matrix_inverse_pth_root_batched(mats, [p]*len(mats))
  best = 1e9
  for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): r, m = K.matrix_inverse_pth_root_batched(mats, [p]*len(mats))
    torch.cuda.synchronize(); best = min(best, (time.perf_counter()-t0)/5*1e3)
  print("   ", name, "%.2f ms" % best, "checksum %.9e" % float(torch.stack(r).double().abs().sum()), flush=True)
  del st, mats, r; torch.cuda.empty_cache()
if "vit" in os.environ.get("PS_DEV_EXTRA", ""):
  w = bench.VitBWorkload(0, 1, dev, None)
  for _ in range(2): w.step()
  best = 1e9
  for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): w.step()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter()-t0)/2*1e3)
  print("    vit_b %.1f ms" % best, flush=True)
'''
combos = [dict(), dict(PS_NEWTON_PIPE="0", PS_NEWTON_BK="16", PS_NEWTON_DEEP="1"), dict(PS_NEWTON_PIPE="0", PS_NEWTON_BK="16", PS_NEWTON_DEEP="0"), dict(), dict(PS_NEWTON_PIPE="0", PS_NEWTON_BK="16", PS_NEWTON_DEEP="1")]
if len(sys.argv) > 1 and sys.argv[1] == "bk":
  combos = [dict(PS_NEWTON_BK=bk, PS_NEWTON_DEEP=deep) for bk, deep in (("32", "1"), ("16", "1"), ("16", "0"), ("32", "0"))]
if len(sys.argv) > 1 and sys.argv[1] == "groups":   # round 6: stream groups of the staged execution (same bits for every count)
  os.environ["PS_DEV_EXTRA"] = "vit"
  combos = [dict(PS_NEWTON_GROUPS=g) for g in ("1", "2", "3", "4", "1", "2")]
if len(sys.argv) > 1 and sys.argv[1] == "grid":   # stage launches capped at g workgroups walking the tile list (descriptor prefetch)
  combos = [dict(), dict(PS_NEWTON_GRID="512"), dict(PS_NEWTON_GRID="1024"), dict(PS_NEWTON_GRID="256"), dict()]
for c in combos:
  env = dict(os.environ, **c)
  print(c, flush=True)
  subprocess.run([sys.executable, "-c", CHILD], env=env, check=False, timeout=300)
