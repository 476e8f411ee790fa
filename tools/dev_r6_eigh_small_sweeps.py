"""Dev (round 6): how the LDS-resident 96 x 96 eigensolver's time depends on its input (the Rayleigh-Ritz matrices of
the FD rounds become nearly diagonal as the subspace converges): random SPD against diagonal + small off-diagonal."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
gen = torch.Generator(device="cpu").manual_seed(5)
n, B = 96, 8
def run(mats, label):
  for _ in range(3): K.eigh_batched(mats)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(20): K.eigh_batched(mats)
  torch.cuda.synchronize()
  print("%-40s %.3f ms per call of %d" % (label, (time.perf_counter() - t0) / 20 * 1e3, B), flush=True)
g = torch.randn((B, n, 4 * n), generator=gen)
run([(x @ x.T).to(dev).contiguous() for x in g], "random Wishart")
for eps in (1e-1, 1e-2, 1e-3, 1e-5, 0.0):
  mats = []
  for j in range(B):
    d = torch.linspace(10.0, 1.0, n) ** 2
    e = torch.randn((n, n), generator=gen) * eps
    mats.append((torch.diag(d) + (e + e.T)).to(dev).contiguous())
  run(mats, "diag(100 .. 1) + %.0e * noise" % eps)
