"""Dev (round 6): the FD sketch update as ONE library call (ps_fd_update_batched_f32) against the step-by-step path
(PS_FD_ONE_CALL=0): bench.fd_cfg5 with 8 factors and with one factor per GPU, and the literal-input parity."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
for mode in ("1", "0", "1", "0"):
  os.environ["PS_FD_ONE_CALL"] = mode
  r8 = bench.fd_cfg5(dev)
  r1 = bench.fd_cfg5(dev, factors=1)
  print("PS_FD_ONE_CALL", mode, "8 factors ms/factor", r8["ms_per_factor_update"], " one factor ms", r1["ms_per_factor_update"], flush=True)
os.environ["PS_FD_ONE_CALL"] = "1"
par = bench.fd_parity_literal(dev, updates=2)
print(json.dumps({k: v for k, v in par.items() if k != "updates"}), flush=True)
for row in par["updates"]:
  print(row, flush=True)
