"""Dev (round 6): stress of the tall-strip team step of quant_strip_kernel (parts of a strip on different workgroups
merge their column maxima and wait for each other): many repetitions of grouped calls with tall matrices, alone and
next to a stream that keeps the chip busy, every result against the oracle's (computed once)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import quantization_oracle as qorc
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
shapes = [(3072, 768), (2048, 2048), (1500, 260), (4096, 64), (1025, 128), (3072, 768), (2500, 1000)]
xs = [np.ascontiguousarray((rng.standard_normal(s) * np.exp(rng.uniform(-4, 4, size=s[1]))).astype(np.float32)) for s in shapes]
ts = [torch.tensor(x, device=dev) for x in xs]
bad = 0
reps = int(os.environ.get("REPS", "300"))
for bits, tq, npdt in ((8, torch.int8, np.int8), (16, torch.int16, np.int16)):
  ref = [qorc.quantize(x, npdt, False) for x in xs]
  refq = [torch.tensor(r[0], device=dev) for r in ref]
  refb = [torch.tensor(np.asarray(r[2], np.float32), device=dev) for r in ref]
  a = torch.randn((4096, 4096), device=dev)
  side = torch.cuda.Stream()
  for rep in range(reps):
    if rep % 3 == 1:
      with torch.cuda.stream(side):
        for _ in range(3): a = torch.nn.functional.normalize(a @ a, dim=0)
    out = K.quantize_grouped(ts, tq, False)
    for i, (q, d, b) in enumerate(out):
      if not (torch.equal(q, refq[i]) and torch.equal(b.view(torch.int32), refb[i].view(torch.int32))):
        bad += 1
        nb = int((b.view(torch.int32) != refb[i].view(torch.int32)).sum())
        print("MISMATCH bits", bits, "rep", rep, "shape", shapes[i], "wrong buckets", nb, "wrong codes", int((q != refq[i]).sum()), flush=True)
  torch.cuda.synchronize()
print("team stress: reps", reps, "x 2 widths, mismatches", bad)
