"""dev (GPU): cfg5 FD update time, 8 factors and 1 factor per GPU, with the per-step Python loop
(PS_FD_ROUND_CALL=0) and with the whole filter in one library call (default)."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda", 0)
for mode in ("0", "1"):
  os.environ["PS_FD_PLANS"] = mode
  os.environ["PS_FD_ROUND_CALL"] = mode
  for factors in (8, 1):
    r = bench.fd_cfg5(dev, factors=factors, updates=4)
    print(f"PS_FD_ROUND_CALL={mode} factors={factors}: ms per factor update {r['ms_per_factor_update']} tail {r['tail_after_updates']}", flush=True)
