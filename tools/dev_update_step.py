"""Every-step cost of update_fn on the ViT-B tree (dev only)."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import precondition_amd as pa
import bench
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
params = [torch.from_numpy((rng.standard_normal(s) * 0.02).astype(np.float32)).to(dev) for s in bench.VIT_B_SHAPES]
grads = [torch.from_numpy((rng.standard_normal(s) * 0.02).astype(np.float32)).to(dev) for s in bench.VIT_B_SHAPES]
kw = {}
if "--quant" in sys.argv:  # int8 momentum + int16 statistics / preconditioners (needs a batch axis)
  import tempfile
  import torch.distributed as dist
  dist.init_process_group("nccl", store=dist.FileStore(os.path.join(tempfile.mkdtemp(), "s"), 1),
                          rank=0, world_size=1)
  kw = dict(best_effort_memory_usage_reduction=True, batch_axis_name=dist.group.WORLD)
if "--donate" in sys.argv:   # the in-place, allocation-free every-step path (plan.DonatedStep)
  kw["donate_state"] = True
opt = pa.distributed_shampoo(0.1, 1024, preconditioning_compute_steps=50, start_preconditioning_step=1, graft_type=pa.GraftingType.RMSPROP_NORMALIZED, **kw)
st = opt.init(params)
torch.cuda.synchronize()
times = []
for t in range(8):
  t0 = time.perf_counter()
  upd, st = opt.update(grads, st, params)
  torch.cuda.synchronize()
  times.append((time.perf_counter() - t0) * 1e3)
print("update ms per step (step 0 includes the root recompute):", [round(x, 1) for x in times])
# throughput without a host sync per step (how a training loop runs it)
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(20):
  upd, st = opt.update(grads, st, params)
t_host = (time.perf_counter() - t0) / 20
torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / 20
print(f"20 unsynchronised steps: host {t_host*1e3:.2f} ms/step (includes back-pressure of a full launch queue), wall {t_all*1e3:.2f} ms/step")
# host cost proper: enqueue time of 2 steps into an EMPTY queue (the launches never block)
hs = []
for rep in range(5):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for t in range(2):
    upd, st = opt.update(grads, st, params)
  hs.append((time.perf_counter() - t0) / 2)
torch.cuda.synchronize()
print(f"host enqueue time, empty queue: {min(hs)*1e3:.2f} ms/step (median {sorted(hs)[2]*1e3:.2f})")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for t in range(3):
  upd, st = opt.update(grads, st, params)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
nbytes = sum(x.numel() * x.element_size() for x in pa.pytree.tree_leaves(st) if isinstance(x, torch.Tensor))
print(f"optimizer state: {nbytes / 2**20:.1f} MiB")
if "--quant" in sys.argv:
  dist.destroy_process_group()
