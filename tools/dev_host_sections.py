"""Dev: wall-clock split of update() on the ViT-B tree (needs the temporary section timers)."""
import os, sys, time, gc
if os.environ.get("NOGC"): gc.disable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import precondition_amd as pa
import bench
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
params = [torch.from_numpy((rng.standard_normal(s) * 0.02).astype(np.float32)).to(dev) for s in bench.VIT_B_SHAPES]
grads = [torch.from_numpy((rng.standard_normal(s) * 0.02).astype(np.float32)).to(dev) for s in bench.VIT_B_SHAPES]
opt = pa.distributed_shampoo(0.1, 1024, preconditioning_compute_steps=1000, start_preconditioning_step=1, graft_type=pa.GraftingType.RMSPROP_NORMALIZED)
st = opt.init(params)
for _ in range(3): upd, st = opt.update(grads, st, params)
torch.cuda.synchronize()
T = getattr(opt.update, "__dict__", {}).get("T")
if T is not None: T.clear()
N = 20
t0 = time.perf_counter()
for _ in range(N): upd, st = opt.update(grads, st, params)
th = (time.perf_counter() - t0) / N
torch.cuda.synchronize()
tg = (time.perf_counter() - t0) / N
print(f"host per step {th*1e3:.2f} ms; incl. final sync {tg*1e3:.2f} ms")
if T:
  for k, v in T.items(): print(f"  {k}: {v/N*1e3:.2f} ms")
