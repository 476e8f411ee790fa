"""Dev (round 6): the FD update of 8 factors as one group / two interleaved groups (PS_FD_GROUPS), child processes."""
import os, subprocess, sys
os.environ["PS_DEV_ENV"] = "1"
CHILD = r'''
import os, sys
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda:0")
r8 = bench.fd_cfg5(dev)
r1 = bench.fd_cfg5(dev, factors=1)
print("8 factors ms/factor", r8["ms_per_factor_update"], " one factor ms", r1["ms_per_factor_update"], flush=True)
'''
for c in (dict(PS_FD_GROUPS="2"), dict(PS_FD_GROUPS="1"), dict(PS_FD_GROUPS="2"), dict(PS_FD_GROUPS="1")):
  out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **c), capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  print(c, (out.stdout.strip().splitlines() or [out.stderr[-600:]])[-1], flush=True)
