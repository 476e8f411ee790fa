"""Dev: which reference cycles does one update() leave behind (they delay freeing device memory
until the cyclic collector runs)."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, gc, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import precondition_amd as pa
import bench
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
shapes = bench.VIT_B_SHAPES[:24]
params = [torch.from_numpy((rng.standard_normal(s) * 0.02).astype(np.float32)).to(dev) for s in shapes]
grads = [torch.from_numpy((rng.standard_normal(s) * 0.02).astype(np.float32)).to(dev) for s in shapes]
opt = pa.distributed_shampoo(0.1, 1024, preconditioning_compute_steps=1000, start_preconditioning_step=1, graft_type=pa.GraftingType.RMSPROP_NORMALIZED)
st = opt.init(params)
for _ in range(3): upd, st = opt.update(grads, st, params)
gc.collect()
gc.disable()
gc.set_debug(gc.DEBUG_SAVEALL)
upd, st = opt.update(grads, st, params)
n = gc.collect()
print("unreachable objects found:", n)
cnt = collections.Counter(type(o).__name__ for o in gc.garbage)
print(cnt.most_common(15))
tens = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
print("tensors in cycles:", len(tens), sum(t.numel() for t in tens) * 4 / 1e6, "MB")
# who refers to the first few non-tensor objects
for o in gc.garbage[:400]:
  if type(o).__name__ in ("function", "cell", "frame", "dict", "list", "tuple") :
    continue
  if isinstance(o, torch.Tensor): continue
  print(type(o), str(o)[:120])
funcs = [o for o in gc.garbage if type(o).__name__ == "function"]
for f in funcs[:20]: print("func", f.__qualname__, f.__code__.co_filename.split("/")[-1], f.__code__.co_firstlineno)
