"""dev (GPU): A/B of the round-4 Newton policy on cfg2 / headline / ViT-B in ONE process.
  careful   no hint: 4 averaged steps + segmented accumulation (CAREFUL kernel, 1 workgroup per CU)
  chain     no hint, accumulation = chain (the round-3 default: 2 workgroups per CU)
  hinted    last step's iteration counts as ps_options.iters_hint (fast path where <= 8)
  + PS_NEWTON_AVG_LPT=0 (dev env) for the tile order of averaged launches.
Usage: python tools/dev_r4_newton.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PS_DEV_ENV"] = "1"
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)


def run(w, label, options, hint):
  w.options = dict(options)
  w.hint = None
  w.compute(); torch.cuda.synchronize()
  if hint:
    w.refresh_hint()
  w.compute(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    w.compute()
  torch.cuda.synchronize()
  ms = (time.perf_counter() - t0) / steps * 1e3
  m = w.metrics.cpu().numpy()
  sm, ln, pm, om = bench.profile_stage_kernel(w)
  print(f"  {label:34s} {ms:8.3f} ms/step  iters {m[:, 1].min():.0f}-{m[:, 1].max():.0f}  avg steps {m[:, 7].mean():.1f}"
        f"  stage {sm:7.3f} ms / {ln} launches  pi {pm:.3f}  other {om:.3f}", flush=True)
  return ms


for name in ("cfg2_256x512_p4", "headline_64x1024_p4"):
  print(name, flush=True)
  w = bench.Workload(name, 0, 1, dev)
  run(w, "careful (no hint)", {}, False)
  run(w, "chain (no hint) = round 3", {"accumulation": "chain"}, False)
  run(w, "hinted (fast)", {}, True)
  os.environ["PS_NEWTON_AVG_LPT"] = "0"
  run(w, "chain, avg tiles in list order", {"accumulation": "chain"}, False)
  run(w, "careful, avg tiles in list order", {}, False)
  os.environ.pop("PS_NEWTON_AVG_LPT")
  print("  parity (hinted):", bench.parity_sample(w, count=2), flush=True)
  del w
  torch.cuda.empty_cache()

print("ViT-B (cfg4)", flush=True)
from precondition_amd import comm  # noqa: E402
vw = bench.VitBWorkload(0, 1, dev, None)
flat = [s for st in vw.stats for s in st]
for label, opts, use_hint in (("careful (no hint)", None, False),
                              ("chain (no hint) = round 3", {"accumulation": "chain"}, False),
                              ("hinted", None, True)):
  hint = None
  for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    roots, met = comm.sharded_inverse_pth_roots(flat, vw.exps, group=None, ownership="lpt", pi_first=True,
                                                iters_hint=hint, hint_in_ownership=False, options=opts)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    if use_hint:
      hint = met[:, 1].cpu().numpy().tolist()
  vw.metrics = met
  par = bench.parity_sample_vit_b(vw, roots, met)
  print(f"  {label:28s} roots {ms:8.2f} ms  max build/oracle err vs f64 {par['max_build_over_oracle_error_vs_f64']:.3f}"
        f"  rel_fro_max vs oracle {par['rel_fro_max']:.2e}  iters equal {par['iteration_and_retry_counts_equal']}", flush=True)
  for r in par["classes"]:
    print(f"      n={r['n']:5d} p={r['p']} iters {r['iters']:.0f}  build {r['build_vs_f64']:.3e}  oracle {r['oracle_vs_f64']:.3e}"
          f"  ratio {r['build_vs_f64'] / r['oracle_vs_f64']:.2f}", flush=True)
