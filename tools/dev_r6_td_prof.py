"""Dev (round 6): 48 eigh roots of 1024 rows on the tridiagonalisation path, statistics of rank-8 gradients after 7
updates from epsilon I (arg "lowrank") or Wishart blocks (arg "wishart"): run under rocprofv3 --kernel-trace --stats."""
import os, sys, time
os.environ.setdefault("PS_DEV_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "lowrank"
n, nb = 1024, 48
gen = torch.Generator(device=dev).manual_seed(5)
mats = []
for b in range(nb):
  if kind == "wishart":
    g = torch.randn((n, 2 * n), generator=gen, device=dev)
    a = g @ g.T
  else:
    a = 1e-6 * torch.eye(n, device=dev)
    for _ in range(7):
      g = (torch.randn((n, 8), generator=gen, device=dev) @ torch.randn((8, 3 * n), generator=gen, device=dev)) * (0.02 / 3)
      a = 0.999 * a + 0.001 * (g @ g.T)
  mats.append(((a + a.T) / 2).contiguous())
torch.cuda.synchronize()
for rep in range(3):
  t0 = time.perf_counter()
  r, m = K.matrix_inverse_pth_root_batched(mats, [4] * nb, eigh=True)
  torch.cuda.synchronize()
  print(f"{kind}: {1e3 * (time.perf_counter() - t0):.1f} ms  sweeps>0: {int((m[:, 5] > 0).sum())}  pi steps mean {float(m[:, 6].mean()):.0f}", flush=True)
