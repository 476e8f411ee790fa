"""Dev: where the host time of an update() goes (wall-clock around the backend calls)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import precondition_amd as pa
from precondition_amd import kernels as K
import bench
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
params = [torch.from_numpy((rng.standard_normal(s) * 0.02).astype(np.float32)).to(dev) for s in bench.VIT_B_SHAPES]
grads = [torch.from_numpy((rng.standard_normal(s) * 0.02).astype(np.float32)).to(dev) for s in bench.VIT_B_SHAPES]
acc = collections.defaultdict(float)
def wrap(name):
  f = getattr(K, name)
  def g(*a, **k):
    t0 = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t0; return r
  setattr(K, name, g)
for n in ("stats_update_grouped", "gemm_grouped", "transform_grads_fused"):
  wrap(n)
# inside the wrappers: library call alone
L = K.lib()
for n in ("ps_stats_update_grouped_f32", "ps_gemm_grouped_f32", "ps_transform_grads_f32"):
  f = getattr(L, n)
  def mk(f, n):
    def g(*a):
      t0 = time.perf_counter(); r = f(*a); acc["C:" + n] += time.perf_counter() - t0; return r
    return g
  setattr(L, n, mk(f, n))
opt = pa.distributed_shampoo(0.1, 1024, preconditioning_compute_steps=1000, start_preconditioning_step=1, graft_type=pa.GraftingType.RMSPROP_NORMALIZED)
st = opt.init(params)
for _ in range(3): upd, st = opt.update(grads, st, params)
torch.cuda.synchronize(); acc.clear()
N = 10
t0 = time.perf_counter()
for _ in range(N):
  upd, st = opt.update(grads, st, params)
  torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / N
print(f"synced step {tot*1e3:.2f} ms")
for k, v in sorted(acc.items()): print(f"  {k}: {v/N*1e3:.2f} ms/step")
acc.clear(); t0 = time.perf_counter()
for _ in range(N): upd, st = opt.update(grads, st, params)
th = (time.perf_counter() - t0) / N
torch.cuda.synchronize()
print(f"unsynced host {th*1e3:.2f} ms")
for k, v in sorted(acc.items()): print(f"  {k}: {v/N*1e3:.2f} ms/step")
