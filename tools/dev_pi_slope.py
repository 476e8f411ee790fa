"""Dev: per-step cost of the power iteration = slope of time over num_iters."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(3)
for nb, n in ((256, 512), (64, 1024), (128, 512), (512, 256)):
  g = torch.randn((nb, n, 2 * n), generator=gen, device=dev)
  mats = list(torch.bmm(g, g.transpose(1, 2)))
  res = {}
  for iters in (20, 100):
    for _ in range(2):
      K.power_iteration_batched(mats, num_iters=iters, error_tolerance=0.0, symmetry="assume")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
      K.power_iteration_batched(mats, num_iters=iters, error_tolerance=0.0, symmetry="assume")
    torch.cuda.synchronize(); res[iters] = (time.perf_counter() - t0) / 5
  print(f"{nb} x {n}: 20 its {res[20]*1e3:.3f} ms, 100 its {res[100]*1e3:.3f} ms, per step {(res[100]-res[20])/80*1e6:.2f} us")
