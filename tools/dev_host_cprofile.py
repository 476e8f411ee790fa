"""Dev: cProfile of the host side of update() on the ViT-B tree."""
import os, sys, cProfile, pstats
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
pr = cProfile.Profile()
pr.enable()
for _ in range(10): upd, st = opt.update(grads, st, params)
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr); ps.sort_stats(sys.argv[1] if len(sys.argv) > 1 else "tottime").print_stats(28)
