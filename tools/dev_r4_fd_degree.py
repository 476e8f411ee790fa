"""dev (GPU): cfg5 factor updates for several caps of the Chebyshev filter degree (PS_FD_DEGREE):
ms per factor update, outer rounds, and the sketch tail (the oracle-compared quantity)."""
import os
import sys
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, json, torch
sys.path.insert(0, %r)
os.environ["PS_DEV_ENV"] = "1"
import bench
from precondition_amd import subspace
orig = subspace.top_eigenpairs_batched
info = []
def wrapped(*a, **k):
  r = orig(*a, **k); info.append((r[3]["outer_iterations"], r[3]["big_gemms"], r[3]["max_residual_rel"])); return r
subspace.top_eigenpairs_batched = wrapped
r = bench.fd_cfg5(torch.device("cuda:0"), updates=4)
print(os.environ.get("PS_FD_DEGREE"), r["ms_per_factor_update"], r["tail_after_updates"], info[-4:])
''' % ROOT
for deg in sys.argv[1:] or ["12", "16", "20", "24"]:
  env = dict(os.environ, PS_FD_DEGREE=deg)
  out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
  print((out.stdout.strip().splitlines() or [out.stderr[-400:]])[-1], flush=True)
