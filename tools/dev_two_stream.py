"""Dev: root the cfg2 batch as two halves driven by two host threads on two streams."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
for name in ("cfg2_256x512_p4", "headline_64x1024_p4"):
  w = bench.Workload(name, 0, 1, dev)
  nb, n, p = w.nb, w.n, w.p
  def run(lo, hi, stream, delay=0.0):
    with torch.cuda.stream(stream):
      if delay: time.sleep(delay)
      K.matrix_inverse_pth_root_batched(list(w.stats[lo:hi].unbind(0)), [p] * (hi - lo), padding_starts=[n] * (hi - lo), out=list(w.roots[lo:hi].unbind(0)))
  s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
  def both(split, delay):
    h = int(nb * split)
    t1 = threading.Thread(target=run, args=(0, h, s1)); t2 = threading.Thread(target=run, args=(h, nb, s2, delay))
    t1.start(); t2.start(); t1.join(); t2.join()
  for _ in range(2): w.compute()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(5): w.compute()
  torch.cuda.synchronize(); base = (time.perf_counter() - t0) / 5
  ref = w.roots.clone()
  print(name, "single call %.2f ms" % (base * 1e3))
  for split, delay in ((0.5, 0.0), (0.5, 0.001), (0.5, 0.002), (0.6, 0.0015)):
    for _ in range(2): both(split, delay)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): both(split, delay)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("  two threads split %.2f delay %.1f ms: %.2f ms  same=%s" % (split, delay * 1e3, dt * 1e3, torch.equal(ref, w.roots)))
