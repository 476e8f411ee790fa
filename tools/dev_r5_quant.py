"""Dev (round 5): quantize / dequantize legs of bench.py alone, fused vs two-pass kernel."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")
import sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
for flat in (sys.argv[1:] or ["1", "0"]):
  os.environ["PS_QUANT_FLAT"] = flat
  r = bench.quant_f3(dev)
  print("flat", flat, json.dumps({k: v for k, v in r.items() if "plan" in k or "prealloc" in k}))
