"""Dev (round 6): the 8-rank share of the ViT-B recompute, staged execution (2 stream groups) against the
persistent execution (PS_NEWTON_PERSISTENT=1: no launch boundaries, so no partly filled tile rounds)."""
import os, subprocess, sys
os.environ["PS_DEV_ENV"] = "1"
CHILD = r'''
import os, sys
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda:0")
vw = bench.VitBWorkload(0, 1, dev, None)
for _ in range(2):
  vw.step()
torch.cuda.synchronize()
r = bench.vit_b_rank_share(vw, dev, worlds=(4, 8), reps=5)
print("one rank", r["one_rank_ms"], {w: (v["share_ms"], v["projected_speedup"]) for w, v in r["worlds"].items()}, flush=True)
'''
for c in (dict(), dict(PS_NEWTON_PERSISTENT="1"), dict()):
  out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **c), capture_output=True, text=True, timeout=900)
  print(c, (out.stdout.strip().splitlines() or [out.stderr[-600:]])[-1], flush=True)
