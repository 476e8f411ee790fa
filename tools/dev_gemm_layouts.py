"""Dev: grouped fp32 GEMM throughput per operand-layout pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
def case(nb, m, n, k, ta, tb):
  A = [torch.randn((k, m) if ta else (m, k), device=dev) for _ in range(nb)]
  B = [torch.randn((n, k) if tb else (k, n), device=dev) for _ in range(nb)]
  Cs = [torch.empty(m, n, device=dev) for _ in range(nb)]
  items = [(a, b, c, ta, tb) for a, b, c in zip(A, B, Cs)]
  for _ in range(3): K.gemm_grouped(items)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10): K.gemm_grouped(items)
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 10
  fl = nb * 2.0 * m * n * k
  print(f"nb={nb} m={m} n={n} k={k} ta={ta} tb={tb}: {ms:.3f} ms {fl/ms/1e9:.1f} TF/s ({fl/ms/1e9/157.3:.3f})")
for ta in (False, True):
  for tb in (False, True):
    case(64, 1024, 1024, 1024, ta, tb)
case(128, 1024, 1024, 1024, True, False)
case(128, 1024, 1024, 1024, False, False)
case(72, 1024, 768, 768, True, False)
case(72, 768, 1024, 1024, True, False)
