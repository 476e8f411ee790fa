"""Does splitting the batch into sequential sub-batches cost time? (dev only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from precondition_amd import kernels as K
sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
for name in ("cfg2_256x512_p4", "headline_64x1024_p4"):
  w = bench.Workload(name, 0, 1, dev)
  mats = list(w.stats.unbind(0)); outs = list(w.roots.unbind(0)); nb, n, p = w.nb, w.n, w.p
  def run(parts):
    sz = nb // parts
    for k in range(parts):
      K.matrix_inverse_pth_root_batched(mats[k*sz:(k+1)*sz], [p]*sz, [n]*sz, out=outs[k*sz:(k+1)*sz])
  for parts in (1, 2, 4, 1, 2, 4):
    run(parts); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): run(parts)
    torch.cuda.synchronize()
    print(name, "parts", parts, "ms/step %.3f" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)

  def run_fracs(fracs):
    lo = 0
    for i, f in enumerate(fracs):
      hi = nb if i == len(fracs) - 1 else lo + max(1, int(round(nb * f)))
      K.matrix_inverse_pth_root_batched(mats[lo:hi], [p]*(hi-lo), [n]*(hi-lo), out=outs[lo:hi])
      lo = hi
  for fracs in ((1.0,), (0.5, 0.5), (0.625, 0.375), (0.5, 0.3, 0.2), (0.4, 0.3, 0.2, 0.1), (0.45, 0.3, 0.25)):
    run_fracs(fracs); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): run_fracs(fracs)
    torch.cuda.synchronize()
    print(name, "fracs", fracs, "ms/step %.3f" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)

  def run_pi_first(fracs):
    lam, _ = K.power_iteration_batched(mats, padding_starts=[n] * nb)
    lo = 0
    for i, f in enumerate(fracs):
      hi = nb if i == len(fracs) - 1 else lo + max(1, int(round(nb * f)))
      K.matrix_inverse_pth_root_batched(mats[lo:hi], [p]*(hi-lo), [n]*(hi-lo), out=outs[lo:hi], max_ev=lam[lo:hi])
      lo = hi
  run_fracs((1.0,)); torch.cuda.synchronize(); ref = w.roots.clone()
  for fracs in ((1.0,), (0.5, 0.5), (0.5, 0.3, 0.2), (0.4, 0.3, 0.2, 0.1)):
    run_pi_first(fracs); torch.cuda.synchronize()
    same = torch.equal(ref, w.roots)
    t0 = time.perf_counter()
    for _ in range(5): run_pi_first(fracs)
    torch.cuda.synchronize()
    print(name, "PI-first fracs", fracs, "ms/step %.3f" % ((time.perf_counter() - t0) / 5 * 1e3), "bit-identical", same, flush=True)
