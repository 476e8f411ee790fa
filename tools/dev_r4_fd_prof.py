"""dev (GPU): where the host time of a one-factor FD update goes (cProfile) and the GPU-side launch
count (torch profiler kernel count)."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
from precondition_amd import low_rank
dev = torch.device("cuda", 0)
factors = int(sys.argv[1]) if len(sys.argv) > 1 else 1
d, rank = 4096, 64
gen = torch.Generator(device=dev).manual_seed(64)
prevs = [torch.zeros((d, rank + 2), dtype=torch.float32, device=dev) for _ in range(factors)]
def one():
  global prevs
  grads = [torch.randn((d, d), generator=gen, device=dev, dtype=torch.float32) for _ in range(factors)]
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  calls = [dict(new_grad=low_rank.gram_of_block(grads[f], 0), p=4, rank=rank, ridge_epsilon=1e-6, decay=0.999,
                padding_start=d, prev=prevs[f], new_grad_is_gram=True) for f in range(factors)]
  t1 = time.perf_counter()
  prevs = [r[0] for r in low_rank._fd_update_root_batched(calls)]
  t2 = time.perf_counter()
  torch.cuda.synchronize()
  t3 = time.perf_counter()
  return (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3
for _ in range(2):
  one()
print("gram host ms, update host ms, final sync ms:", [tuple(round(x, 2) for x in one()) for _ in range(3)])
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
  one()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
st.sort_stats("tottime").print_stats(18)
