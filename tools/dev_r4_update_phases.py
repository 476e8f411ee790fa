"""dev (GPU): GPU time of the phases of an every-step update() on the ViT-B tree, by events around the
plan's calls (no profiler)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import precondition_amd as pa
from precondition_amd import plan as P
import bench
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
params = [torch.from_numpy((rng.standard_normal(s) * 0.02).astype(np.float32)).to(dev) for s in bench.VIT_B_SHAPES]
grads = [torch.from_numpy((rng.standard_normal(s) * 0.02).astype(np.float32)).to(dev) for s in bench.VIT_B_SHAPES]
acc = {"stats": [], "apply": []}
def wrap(name, key):
  orig = getattr(P.TreePlan, name)
  def f(self, *a, **k):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(self, *a, **k); e1.record(); torch.cuda.synchronize()
    acc[key].append(e0.elapsed_time(e1)); return r
  setattr(P.TreePlan, name, f)
wrap("stats_update", "stats"); wrap("apply_preconditioners", "apply")
opt = pa.distributed_shampoo(0.1, 1024, preconditioning_compute_steps=50, start_preconditioning_step=1, graft_type=pa.GraftingType.RMSPROP_NORMALIZED)
st = opt.init(params)
for t in range(8):
  upd, st = opt.update(grads, st, params)
torch.cuda.synchronize()
print("stats ms", [round(x, 3) for x in acc["stats"][2:]])
print("apply (A + B) ms", [round(x, 3) for x in acc["apply"][2:]])
