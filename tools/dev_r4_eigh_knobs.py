"""dev (GPU): cfg3 eigh root (64 x 2048^2) for a few settings of the developer switches (PS_EIGH_*)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = "import runpy, sys; sys.argv = ['x']; runpy.run_path(%r)" % os.path.join(ROOT, "tools", "dev_eigh_one.py")
settings = [dict(kv.split("=") for kv in a.split(",") if "=" in a) for a in sys.argv[1:]] or [{}]
for env in settings:
  e = dict(os.environ, PS_DEV_ENV="1", **env)
  r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
  print(env, (r.stdout.strip().splitlines() or [r.stderr[-200:]])[-1], flush=True)
