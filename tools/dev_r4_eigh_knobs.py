"""dev (GPU): cfg3 eigh root (64 x 2048^2) for a few settings of the pivot's inner-sweep rule."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = "import runpy, sys; sys.argv = ['x']; runpy.run_path(%r)" % os.path.join(ROOT, "tools", "dev_eigh_one.py")
for env in ({}, {"PS_EIGH_CJ_ONE_BELOW": "0.05"}, {"PS_EIGH_CJ_ONE_BELOW": "0.2"}, {"PS_EIGH_CJ_ONE_BELOW": "0.4"},
            {"PS_EIGH_CJ_INNER": "1"}, {"PS_EIGH_CJ_INNER": "3", "PS_EIGH_CJ_ONE_BELOW": "0.03"}, {"PS_EIGH_CJ_STREAMS": "1"}):
  e = dict(os.environ, PS_DEV_ENV="1", **env)
  r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
  print(env, (r.stdout.strip().splitlines() or [r.stderr[-200:]])[-1], flush=True)
