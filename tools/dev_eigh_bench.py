import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
for nb, n in ((64, 2048), (64, 1024), (256, 512)):
  gen = torch.Generator(device=dev).manual_seed(n)
  stats = torch.zeros((nb, n, n), device=dev)
  for b0 in range(0, nb, 8):
    g = torch.randn((8, n, 2 * n), generator=gen, device=dev)
    K.stats_update_grouped([(g[i], 0, stats[b0+i], stats[b0+i]) for i in range(8)], 0.0, 1.0)
  torch.cuda.synchronize()
  roots = torch.empty_like(stats)
  for rep in range(2):
    t0 = time.perf_counter()
    _, m = K.matrix_inverse_pth_root_batched(list(stats.unbind(0)), [2]*nb, [n]*nb, eigh=True, out=list(roots.unbind(0)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
  m = m.cpu().numpy()
  conv = (6 + 2/3 + 4) * n**3 * nb
  print(f"eigh {nb}x{n}: {dt*1e3:.1f} ms, sweeps {m[:,5].min()}-{m[:,5].max()}, err max {m[:,0].max():.2e}, conventional {conv/dt/1e12:.2f} TFLOP/s")
  t0 = time.perf_counter()
  _, m2 = K.matrix_inverse_pth_root_batched(list(stats.unbind(0)), [2]*nb, [n]*nb, out=list(roots.unbind(0)))
  torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
  print(f"   newton p=2 same input: {dt2*1e3:.1f} ms iters {m2[:,1].max().item()}")
