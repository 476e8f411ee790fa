"""A/B of PS_NEWTON_BK in separate processes on the same box (dev only)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(2):
  for bk in ("16", "32"):
    env = dict(os.environ, PS_NEWTON_BK=bk)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "8", "--warmup", "2"],
                         env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("BK", bk, "cfg2 value", d["value"], "ms", d["ms_per_step"], "stage TF", d["roofline"]["achieved"],
          "| 1024 value", d["headline_1024"]["value"], "ms", d["headline_1024"]["ms_per_step"], "stage TF", d["headline_1024"]["roofline_stage_kernel_tflops"], flush=True)
