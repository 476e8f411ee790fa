import sys; sys.path.insert(0, ".")
import torch, bench, numpy as np, time
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
for name in ("cfg2_256x512_p4", "headline_64x1024_p4"):
  st, p = bench.make_blocks(name, 0, dev); mats = list(st.unbind(0))
  for _ in range(2): r, m = K.matrix_inverse_pth_root_batched(mats, [p]*len(mats))
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(5): r, m = K.matrix_inverse_pth_root_batched(mats, [p]*len(mats))
  torch.cuda.synchronize()
  print(name, "adaptive default: %.2f ms" % ((time.perf_counter()-t0)/5*1e3), "avg steps", m[:,7].min().item(), m[:,7].max().item(), "iters", m[:,1].max().item())
  del st, mats; torch.cuda.empty_cache()
