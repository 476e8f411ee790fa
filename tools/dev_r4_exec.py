"""dev (GPU): staged vs persistent execution of the Newton loop on cfg2 / headline (hinted), ms per step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PS_DEV_ENV"] = "1"
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
for name in ("cfg2_256x512_p4", "headline_64x1024_p4"):
  w = bench.Workload(name, 0, 1, dev)
  for mode in ("staged", "persistent", "staged", "persistent"):
    w.options = {"execution": mode}
    w.hint = None
    for _ in range(2):
      w.compute()
    w.refresh_hint()
    for _ in range(4):
      w.compute()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
      w.compute()
    torch.cuda.synchronize()
    print(f"{name} {mode:10s} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
  del w
  torch.cuda.empty_cache()
