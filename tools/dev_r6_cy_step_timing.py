"""Dev (round 6): one fused Chebyshev filter step (fd_cy_step_kernel) of 8 x 4096^2 by block width b: the
covariance stream (HBM, 4 B per element) is the same for every b, the iterate planes re-read from L2 grow with b --
how much of the step's time is theirs?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
B, n = 8, 4096
c16 = [K.TiledBf16(torch.zeros((n * n,), dtype=torch.bfloat16, device=dev), torch.zeros((n * n,), dtype=torch.bfloat16, device=dev), n, n, 2)
       for _ in range(B)]
for b in (32, 64, 96):
  y = torch.zeros((B, n, b), device=dev); yp = torch.zeros_like(y); yn = torch.zeros_like(y)
  yt = (torch.zeros((B * n * b,), dtype=torch.bfloat16, device=dev), torch.zeros((B * n * b,), dtype=torch.bfloat16, device=dev))
  nt = (torch.zeros_like(yt[0]), torch.zeros_like(yt[1]))
  params = torch.tensor([[1.0, 2.0, 0.5, 64.0]] * B, device=dev)
  for _ in range(5): K.fd_cy_step(c16, yt, y, yp, yn, nt, params, 3)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(50): K.fd_cy_step(c16, yt, y, yp, yn, nt, params, 3)
  torch.cuda.synchronize()
  t = (time.perf_counter() - t0) / 50
  hbm = B * n * n * 4
  print("b %3d  %.1f us per step   covariance stream %.2f TB/s" % (b, t * 1e6, hbm / t / 1e12), flush=True)
