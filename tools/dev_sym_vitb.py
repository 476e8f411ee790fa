import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from precondition_amd import comm
dev = torch.device("cuda:0")
vw = bench.VitBWorkload(0, 1, dev, None)
vw.stats_step()
flat = [s for st in vw.stats for s in st]
bad = [(i, tuple(s.shape), vw.exps[i], float((s - s.T).abs().max()), float(s.abs().max())) for i, s in enumerate(flat) if not torch.equal(s, s.T)]
print("asymmetric statistics:", len(bad), bad[:6])
roots, metrics = comm.sharded_inverse_pth_roots(flat, vw.exps, group=None, ownership="lpt", pi_first=True)
badr = [(i, tuple(h.shape), vw.exps[i], float((h - h.T).abs().max()), float(h.abs().max())) for i, h in enumerate(roots) if not torch.equal(h, h.T)]
print("asymmetric roots:", len(badr), badr[:6])
