"""Dev: the one-sided Cholesky-Jacobi eigh root against float64 / LAPACK float32 and the old
two-sided path; timing of the cfg3 batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
dev = torch.device("cuda:0")
f = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm(y)


def make(kind, n, seed=0):
  rng = np.random.default_rng(seed)
  if kind == "wishart2":
    g = rng.standard_normal((n, 2 * n)).astype(np.float32); return g @ g.T
  if kind == "rankdef":
    g = rng.standard_normal((n, n // 4)).astype(np.float32); return g @ g.T
  cond = float(kind[6:])
  q, _ = np.linalg.qr(rng.standard_normal((n, n)))
  w = cond ** (-np.arange(n) / (n - 1.0))
  a = (q * w) @ q.T
  return (0.5 * (a + a.T)).astype(np.float32)


def check(kind, n, p=2, modes=("1", "0")):
  a = make(kind, n)
  a_d = torch.tensor(a, device=dev)
  out = []
  w, v = np.linalg.eigh(a.astype(np.float64))
  for cj in modes:
    os.environ["PS_EIGH_CJ"] = cj
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r, m = K.matrix_inverse_pth_root_batched([a_d], [p], eigh=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    m = m.cpu().numpy()
    h = r[0].cpu().numpy()
    eps = 1e-6 * float(w.max())
    h64 = (v * np.maximum(w + eps, eps) ** (-1.0 / p)) @ v.T
    w32, v32 = np.linalg.eigh(a + np.float32(eps) * np.eye(n, dtype=np.float32))
    h32 = (v32 * np.maximum(w32, np.float32(eps)) ** np.float32(-1.0 / p)) @ v32.T
    out.append("cj=%s: vs f64 %.2e (lapack32 %.2e) err-metric %.2e sweeps %d  %.1f ms" % (
        cj, f(h, h64), f(h32, h64), m[0, 0], m[0, 5], dt * 1e3))
  print(kind, n, "p", p, " | ".join(out), flush=True)


if __name__ == "__main__":
  for kind, n in (("wishart2", 256), ("wishart2", 384), ("wishart2", 1000), ("graded1e6", 512),
                  ("rankdef", 768), ("wishart2", 2048), ("rankdef", 2048)):
    check(kind, n)
  check("wishart2", 512, p=4)
  if "nobench" not in sys.argv:
    import bench
    for cj in ("1", "0"):
      os.environ["PS_EIGH_CJ"] = cj
      ew = bench.Workload("eigh_cfg3_64x2048_p2", 0, 1, dev)
      ew.step(); torch.cuda.synchronize()
      t0 = time.perf_counter(); ew.step(); torch.cuda.synchronize()
      print("cfg3 cj=%s: %.1f ms, sweeps %s, err metric max %.2e" % (
          cj, (time.perf_counter() - t0) * 1e3, float(ew.metrics[:, 5].max()),
          float(ew.metrics[:, 0].max())), flush=True)
      del ew; torch.cuda.empty_cache()
