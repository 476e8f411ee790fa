"""Does splitting the batch into sequential sub-batches cost time? (dev only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from precondition_amd import kernels as K
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
for name in ("cfg2_256x512_p4", "headline_64x1024_p4"):
  w = bench.Workload(name, 0, 1, dev)
  mats = list(w.stats.unbind(0)); outs = list(w.roots.unbind(0)); nb, n, p = w.nb, w.n, w.p
  def run(parts):
    sz = nb // parts
    for k in range(parts):
      K.matrix_inverse_pth_root_batched(mats[k*sz:(k+1)*sz], [p]*sz, [n]*sz, out=outs[k*sz:(k+1)*sz])
  for parts in (1, 2, 4, 1, 2, 4):
    run(parts); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): run(parts)
    torch.cuda.synchronize()
    print(name, "parts", parts, "ms/step %.3f" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
