"""Dev (round 6): quantize kernels A/B -- register strips (one read) / two-pass, bench.quant_f3 in child
processes (the PS_* developer switches are read once per process)."""
import os, subprocess, sys
os.environ["PS_DEV_ENV"] = "1"
combos = [dict(PS_QUANT_STRIP="1"), dict(PS_QUANT_STRIP="0"), dict(PS_QUANT_STRIP="1")]
for c in combos:
  env = dict(os.environ, **c)
  out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, '.'); import torch, bench; r = bench.quant_f3(torch.device('cuda:0')); print({k: v for k, v in r.items() if 'plan' in k})"],
                       env=env, capture_output=True, text=True, timeout=600)
  print(c, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:], flush=True)
