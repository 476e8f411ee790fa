"""Dev: time the grouped statistics launch on the ViT-B tree (bench.py's VitBWorkload).
Under rocprofv3 --kernel-trace the dispatches are told apart by their grid sizes."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from precondition_amd import kernels as K

dev = torch.device("cuda:0")
w = bench.VitBWorkload(0, 1, dev, None)
items = []
for pc, g, st in zip(w.pcs, w.grads, w.stats):
  items.extend(pc.statistics_update_items(st, g, st))
def run(sel, name, reps=10):
  for _ in range(2):
    K.stats_update_grouped(sel, 0.999, 0.001)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  e0.record()
  for _ in range(reps):
    K.stats_update_grouped(sel, 0.999, 0.001)
  e1.record()
  torch.cuda.synchronize()
  fl = 0.0
  for g, axis, _, _ in sel:
    d = g.shape[axis]; t = (d + 127) // 128
    fl += 2.0 * d * g.numel() * (t + 1) / (2.0 * t)
  ms = e0.elapsed_time(e1) / reps
  print(f"{name}: {len(sel)} stats, wall {(time.perf_counter()-t0)/reps*1e3:.3f} ms, "
        f"events {ms:.3f} ms, executed {fl/ms/1e9:.1f} TF/s ({fl/ms/1e9/157.3:.3f})")

def kdim(it):
  g, axis = it[0], it[1]
  return g.numel() // g.shape[axis]

run(items, "all")
run([it for it in items if kdim(it) >= 128], "matrix blocks")
run([it for it in items if kdim(it) < 128], "vectors (k=1)")
run([it for it in items if kdim(it) >= 128 and it[1] == 0], "matrix axis 0")
run([it for it in items if kdim(it) >= 128 and it[1] == 1], "matrix axis 1")
