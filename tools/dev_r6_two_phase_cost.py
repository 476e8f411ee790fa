"""Dev (round 6): what the two-phase step of the N > 1 path costs on one GPU (one-rank RCCL group): the whole batch
in one call / power iteration once + two root calls / the same + the asynchronous all-gathers (bench.Workload.step).
Run under: python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 tools/dev_r6_two_phase_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
from precondition_amd import kernels as K
dist.init_process_group("nccl", device_id=torch.device("cuda:0")) if False else dist.init_process_group("nccl")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_256x512_p4"
w = bench.Workload(name, 0, 1, dev, True)
def t(fn, reps=10):
  for _ in range(3): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(reps): fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
w.compute(); w.refresh_hint()
print("one call (hinted)                 %.2f ms" % t(w.compute), flush=True)
def two_calls():
  h = w.split_point()
  lam, _ = K.power_iteration_batched(list(w.stats.unbind(0)), padding_starts=[w.n] * w.nb)
  for lo, hi in ((0, h), (h, w.nb)):
    w._roots(lo, hi, lam[lo:hi])
print("split point", w.split_point(), "of", w.nb)
print("PI once + two root calls          %.2f ms" % t(two_calls), flush=True)
def pi_only():
  K.power_iteration_batched(list(w.stats.unbind(0)), padding_starts=[w.n] * w.nb)
print("  power iteration alone           %.2f ms" % t(pi_only), flush=True)
h = w.split_point()
lam, _ = K.power_iteration_batched(list(w.stats.unbind(0)), padding_starts=[w.n] * w.nb)
print("  roots of part 1 (%d blocks)     %.2f ms" % (h, t(lambda: w._roots(0, h, lam[0:h]))), flush=True)
print("  roots of part 2 (%d blocks)     %.2f ms" % (w.nb - h, t(lambda: w._roots(h, w.nb, lam[h:w.nb]))), flush=True)
w.compute(); torch.cuda.synchronize(); ref = w.roots.clone(); refm = w.metrics.clone()
print("two-phase step, parts side by side %.2f ms" % t(w.step), flush=True)
torch.cuda.synchronize()
print("  roots bit-identical to the one call:", bool(torch.equal(ref, w.roots)), " metrics:", bool(torch.equal(refm, w.metrics)),
      " gathered order:", w.check_gathered_order(0), flush=True)
os.environ["PS_BENCH_SEQUENTIAL_PARTS"] = "1"
print("two-phase step, parts one by one   %.2f ms" % t(w.step), flush=True)
del os.environ["PS_BENCH_SEQUENTIAL_PARTS"]
print("two-phase step, parts side by side %.2f ms" % t(w.step), flush=True)
for hh in (96, 128, 153, 160, 179, 192, 204, 218, 230):
  t1 = t(lambda: w._roots(0, hh, lam[0:hh]), 6); t2 = t(lambda: w._roots(hh, w.nb, lam[hh:w.nb]), 6)
  print("  split %3d / %3d: %.2f + %.2f = %.2f ms" % (hh, w.nb - hh, t1, t2, t1 + t2), flush=True)
dist.destroy_process_group()
