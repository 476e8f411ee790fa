"""dev (GPU): the two products of the preconditioner application on ViT-B-like blocks in the four
operand layouts of gemm_grouped (transa, transb), as uniform batches: is the MC x MC form
(g^T P as the optimizer issues it) the slow one?"""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from precondition_amd import kernels as K
dev = torch.device("cuda", 0)
def bench(items, flops, reps=10):
  K.gemm_grouped(items); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  plan = K.GemmPlan(items)
  plan.launch(); torch.cuda.synchronize()
  e0.record()
  for _ in range(reps): plan.launch()
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / reps
  return ms, flops / ms / 1e9 / 157.3
nb = 144
for (m, n) in ((768, 1024), (1024, 768), (768, 768)):
  g = torch.randn((nb, m, n), device=dev)
  pl = torch.randn((nb, m, m), device=dev); pl = pl + pl.transpose(1, 2)
  pr = torch.randn((nb, n, n), device=dev); pr = pr + pr.transpose(1, 2)
  x = torch.empty((nb, n, m), device=dev); xt = torch.empty((nb, m, n), device=dev)
  y = torch.empty((nb, m, n), device=dev)
  fa = 2.0 * nb * m * m * n; fb = 2.0 * nb * m * n * n
  # as issued today: X = g^T P_L (transa), Y = X^T P_R (transa)
  a_ms, a_fr = bench([(g[i], pl[i], x[i], True, False) for i in range(nb)], fa)
  b_ms, b_fr = bench([(x[i], pr[i], y[i], True, False) for i in range(nb)], fb)
  # transposed formulation: X^T = P_L g (no trans), Y = X^T P_R (no trans)
  c_ms, c_fr = bench([(pl[i], g[i], xt[i], False, False) for i in range(nb)], fa)
  d_ms, d_fr = bench([(xt[i], pr[i], y[i], False, False) for i in range(nb)], fb)
  ok = torch.equal(xt, x.transpose(1, 2))
  print(f"block {m}x{n}: g^T P_L {a_ms:.3f} ms ({a_fr:.2f})  X^T P_R {b_ms:.3f} ({b_fr:.2f}) | P_L g {c_ms:.3f} ({c_fr:.2f})  Xt P_R {d_ms:.3f} ({d_fr:.2f})  bit-identical X: {ok}", flush=True)
