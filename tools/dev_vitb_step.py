"""Dev: the ViT-B full recompute step (bench.py VitBWorkload) for profiling."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
w = bench.VitBWorkload(0, 1, dev, None)
w.step(); torch.cuda.synchronize()
w.refresh_hint()   # last recompute's iteration counts (what the optimizer passes: ps_options.iters_hint)
w.step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
  w.step()
torch.cuda.synchronize()
print(f"ViT-B recompute {(time.perf_counter()-t0)/3*1e3:.2f} ms per step")
m = w.metrics.cpu().numpy()
import numpy as np
print("iters histogram", np.bincount(m[:, 1].astype(int)), "retries", m[:, 4].max(), "total_iters max", m[:, 5].max())
