"""Dev: statistics kernel on uniform synthetic blocks (compare with the Newton product shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from precondition_amd import kernels as K

dev = torch.device("cuda:0")
def case(nb, m, n, axis, ld=None):
  ld = ld or n
  gs = [torch.randn(m, ld, device=dev)[:, :n] for _ in range(nb)]
  d = m if axis == 0 else n
  st = [torch.zeros(d, d, device=dev) for _ in range(nb)]
  items = [(g, axis, s, s) for g, s in zip(gs, st)]
  for _ in range(3):
    K.stats_update_grouped(items, 0.999, 0.001)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  reps = 10
  for _ in range(reps):
    K.stats_update_grouped(items, 0.999, 0.001)
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / reps
  t = (d + 127) // 128
  fl = nb * 2.0 * d * m * n * (t + 1) / (2.0 * t)
  print(f"nb={nb} [{m}x{n}] ld={ld} axis={axis}: {ms:.3f} ms executed {fl/ms/1e9:.1f} TF/s ({fl/ms/1e9/157.3:.3f})")

case(64, 1024, 1024, 0)
case(64, 1024, 1024, 1)
case(64, 768, 768, 0)
case(64, 768, 768, 1)
case(36, 768, 1024, 0, 3072)
case(36, 768, 1024, 1, 3072)
case(64, 1024, 4096, 0)
case(64, 4096, 1024, 1)
case(128, 1024, 1024, 0)
