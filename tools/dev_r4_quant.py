"""dev (GPU): quantize / dequantize of the ViT-B statistics (bench.quant_f3) -- wall clock per call vs the
kernels' own time (run under rocprofv3 --kernel-trace --stats for the latter)."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")
import sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
print(json.dumps({k: v for k, v in bench.quant_f3(torch.device("cuda:0")).items() if "ms" in k or "GBps" in k}))
