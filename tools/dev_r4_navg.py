"""dev (GPU): error of the ViT-B roots against float64 (relative to the oracle's) and root time for
0 ... 4 averaged M updates, with segmented accumulation."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")
import sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from precondition_amd import comm
dev = torch.device("cuda", 0)
vw = bench.VitBWorkload(0, 1, dev, None)
for _ in range(3): vw.stats_step()      # the harder state bench.py's parity sample sees
flat = [s for st in vw.stats for s in st]
for navg in (4, 3, 2, 1, 0):
  for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    roots, met = comm.sharded_inverse_pth_roots(flat, vw.exps, group=None, ownership="lpt", options={"averaged_steps": navg})
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
  vw.metrics = met
  par = bench.parity_sample_vit_b(vw, roots, met)
  print(f"navg {navg}: roots {ms:7.2f} ms  max build/oracle {par['max_build_over_oracle_error_vs_f64']:.3f}  " +
        "  ".join(f"{r['n']}/p{r['p']}:{r['build_vs_f64'] / r['oracle_vs_f64']:.2f}" for r in par["classes"]), flush=True)
