"""Dev: HBM roofline of the quantize / dequantize kernels on the ViT-B state (cfg4)."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from precondition_amd import kernels as K
from precondition_amd.blocking import Preconditioner
from bench import VIT_B_SHAPES
dev = torch.device("cuda:0")
stats, moms = [], []
for shape in VIT_B_SHAPES:
  p = torch.randn(shape, device=dev)
  if len(shape) > 1:
    moms.append(p)
  pc = Preconditioner(p, 1024, 4096, True)
  for s in pc.shapes_for_preconditioners():
    g = torch.randn(s[0], 64, device=dev)
    stats.append(g @ g.T)
ne = sum(s.numel() for s in stats)
nm = sum(m.numel() for m in moms)
def timeit(fn, reps=5):
  fn(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps): out = fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / reps, out
tq, triples = timeit(lambda: K.quantize_grouped(stats, torch.int16, True))
td, _ = timeit(lambda: K.dequantize_grouped(triples))
print(f"int16 stats: {len(stats)} matrices, {ne/1e6:.1f} M elements: quantize {tq*1e3:.2f} ms = {ne*6/tq/1e9:.0f} GB/s algorithmic (4B read + 2B write), dequantize {td*1e3:.2f} ms = {ne*6/td/1e9:.0f} GB/s")
tq, triples = timeit(lambda: K.quantize_grouped(moms, torch.int8, False))
td, _ = timeit(lambda: K.dequantize_grouped(triples))
print(f"int8 momentum: {len(moms)} tensors, {nm/1e6:.1f} M elements: quantize {tq*1e3:.2f} ms = {nm*5/tq/1e9:.0f} GB/s, dequantize {td*1e3:.2f} ms = {nm*5/td/1e9:.0f} GB/s")
big = [torch.randn(8192, 8192, device=dev) for _ in range(8)]
nb = sum(b.numel() for b in big)
tq, triples = timeit(lambda: K.quantize_grouped(big, torch.int16, False))
td, _ = timeit(lambda: K.dequantize_grouped(triples))
print(f"8 x 8192^2 int16: quantize {tq*1e3:.2f} ms = {nb*6/tq/1e9:.0f} GB/s, dequantize {td*1e3:.2f} ms = {nb*6/td/1e9:.0f} GB/s")
