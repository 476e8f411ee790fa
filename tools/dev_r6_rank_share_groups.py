"""Dev (round 6): the 8-rank share of the ViT-B recompute (bench.vit_b_rank_share) with 1 ... 4 stream groups."""
import os, sys
os.environ.setdefault("PS_DEV_ENV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
vw = bench.VitBWorkload(0, 1, dev, None)
for _ in range(2):
  vw.step()
torch.cuda.synchronize()
for g in ("1", "2", "3", "4", "1", "2", "3", "4"):
  os.environ["PS_NEWTON_GROUPS"] = g
  r = bench.vit_b_rank_share(vw, dev, worlds=(4, 8), reps=5)
  print("groups", g, "one rank", r["one_rank_ms"], {w: (v["share_ms"], v["projected_speedup"]) for w, v in r["worlds"].items()}, flush=True)
