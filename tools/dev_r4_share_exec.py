"""dev (GPU): the most loaded rank's share of the ViT-B recompute for worlds 8 / 4 with the staged and the
persistent execution of the Newton loop (few tile rounds per stage launch: does the dataflow kernel,
which has no launch boundaries between the products of a step, do better there?)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PS_DEV_ENV"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from precondition_amd import comm, kernels as K  # noqa: E402

dev = torch.device("cuda", 0)
vw = bench.VitBWorkload(0, 1, dev, None)
vw.step(); torch.cuda.synchronize(); vw.refresh_hint(); vw.step(); torch.cuda.synchronize()
flat = [s_ for st_ in vw.stats for s_ in st_]
sizes = [int(s_.shape[0]) for s_ in flat]
hint = vw.hint
cost = comm.block_costs(sizes, vw.exps, hint)
for w in (8, 4, 2):
  owner = comm.ownership_table(sizes, vw.exps, w, "lpt", hint)
  load = [sum(c for c, o in zip(cost, owner) if o == r) for r in range(w)]
  crit = int(np.argmax(load))
  idx = [i for i, o in enumerate(owner) if o == crit]
  mats = [flat[i] for i in idx]
  for mode in ("staged", "persistent"):
    opts = {"iters_hint": np.asarray([hint[i] for i in idx], np.float32), "execution": mode}
    ms = []
    for _ in range(4):
      torch.cuda.synchronize(); t0 = time.perf_counter()
      _, m = K.matrix_inverse_pth_root_batched(mats, [vw.exps[i] for i in idx], padding_starts=[sizes[i] for i in idx], options=opts)
      torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    print(f"world {w}: {len(idx)} statistics, {mode:10s} roots {np.median(ms):7.2f} ms", flush=True)
