"""dev (GPU): start-delay ("stagger") of half of the first resident workgroups of a product launch:
PS_NEWTON_STAGGER = (mode << 16) | microseconds; ms per step of cfg2 / headline for several settings."""
import os
import sys
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
import bench
dev = torch.device("cuda", 0)
out = []
for name in ("cfg2_256x512_p4", "headline_64x1024_p4"):
  w = bench.Workload(name, 0, 1, dev)
  for _ in range(3): w.compute()
  w.refresh_hint()
  for _ in range(5): w.compute()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(20): w.compute()
  torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 20 * 1e3)
  del w; torch.cuda.empty_cache()
vw = bench.VitBWorkload(0, 1, dev, None)
vw.step(); torch.cuda.synchronize(); vw.refresh_hint(); vw.step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): vw.step()
torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 3 * 1e3)
print("stagger %%s: cfg2 %%.3f ms  headline %%.3f ms  ViT-B %%.2f ms" %% (os.environ.get("PS_NEWTON_STAGGER"), out[0], out[1], out[2]))
''' % ROOT
for setting in sys.argv[1:]:
  mode, us = setting.split(":")
  env = dict(os.environ, PS_DEV_ENV="1", PS_NEWTON_STAGGER=str((int(mode) << 16) | int(us)))
  r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
  print(setting, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
