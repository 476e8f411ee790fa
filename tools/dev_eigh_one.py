import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
"""dev (GPU): ONE cfg3 eigh root call (64 x 2048^2) after a warm-up call, for a kernel timeline."""
import sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
nb, n = 64, 2048
gen = torch.Generator(device=dev).manual_seed(n)
stats = torch.zeros((nb, n, n), device=dev)
for b0 in range(0, nb, 8):
  g = torch.randn((8, n, 2 * n), generator=gen, device=dev)
  K.stats_update_grouped([(g[i], 0, stats[b0 + i], stats[b0 + i]) for i in range(8)], 0.0, 1.0)
torch.cuda.synchronize()
roots = torch.empty_like(stats)
for rep in range(2):
  t0 = time.perf_counter()
  _, m = K.matrix_inverse_pth_root_batched(list(stats.unbind(0)), [2] * nb, [n] * nb, eigh=True, out=list(roots.unbind(0)))
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
  print(f"eigh {nb}x{n}: {dt * 1e3:.1f} ms", flush=True)
