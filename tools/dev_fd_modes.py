import sys, os, json, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for mode in ("f32", "bf16x3", "bf16"):
  os.environ["PS_FD_FILTER"] = mode
  r = bench.fd_cfg5(torch.device("cuda:0"))
  print(mode, r["ms_per_factor_update"], r["tail_after_updates"])
