"""Dev (round 6): bench.fd_cfg5 with ONE factor (the 8-GPU shape of BASELINE configs[4]) alone, for
rocprofv3 --kernel-trace --stats: where the 9 ms of a one-factor update go."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
r1 = bench.fd_cfg5(dev, factors=1)
print("one factor ms", r1["ms_per_factor_update"], flush=True)
