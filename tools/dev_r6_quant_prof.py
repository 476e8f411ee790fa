"""Dev (round 6): bench.quant_f3 alone (for rocprofv3 --kernel-trace --stats: kernel durations of the quantized-state
kernels next to the wall clock the bench reports)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
r = bench.quant_f3(torch.device("cuda:0"))
print({k: v for k, v in r.items() if "plan" in k})
