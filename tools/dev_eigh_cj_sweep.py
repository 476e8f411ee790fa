"""Dev: cfg3 step time of the Cholesky-Jacobi eigh path over its tuning switches."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
dev = torch.device("cuda:0")
ew = bench.Workload("eigh_cfg3_64x2048_p2", 0, 1, dev)
for cfg in sys.argv[1:] or ["default"]:
  env = dict(kv.split("=") for kv in cfg.split(",") if "=" in kv)
  for k, v in env.items():
    os.environ[k] = v
  ew.step(); torch.cuda.synchronize()
  t0 = time.perf_counter(); ew.step(); torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  m = ew.metrics.cpu().numpy()
  # accuracy sample: block 0 against float64
  a = ew.stats[0].cpu().numpy().astype(np.float64)
  w, v = np.linalg.eigh(a)
  eps = 1e-6 * w.max()
  h64 = (v * (w + eps) ** -0.5) @ v.T
  h = ew.roots[0].cpu().numpy()
  print("%-60s %.1f ms sweeps %d-%d errmetric %.2e  root vs f64 %.2e  checksum %.12e" % (
      cfg, dt * 1e3, m[:, 5].min(), m[:, 5].max(), m[:, 0].max(),
      np.linalg.norm(h - h64) / np.linalg.norm(h64),
      float(sum(r.double().abs().sum() for r in ew.roots))), flush=True)
  for k in env:
    os.environ.pop(k)
