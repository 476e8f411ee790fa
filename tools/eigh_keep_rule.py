"""Round 6: where does each eigh solver of the build stand against the reference's TRUE arithmetic?

The reference's eigh root (DS:943-1030) runs LAPACK ``ssyevd`` in float32 (jax_enable_x64 off,
DS:35-38).  For statistics families Shampoo produces (graded spectra, rank-deficient + ridge,
low-rank gradient accumulation, Wishart) and n = 169 ... 2048 this prints, per case, the root error
against the float64 closed form of the same float32 matrix (oracle.eigh_root_float64):

    e_ssyevd      the reference's arithmetic (oracle.matrix_inverse_pth_root_eigh, lapack="f32")
    e_f64lapack   NumPy's float64-internal eigh (what rounds 1-5 took for the reference)
    e_tridiagonal the build's fast path kept unconditionally (eigh_solver="tridiagonal")
    e_one_sided   the build's Jacobi solver on the Cholesky factor (eigh_solver="one_sided")
    e_auto        the default rule

and the reference's error metric (PS_M_ERROR = max|U^T D U - diag(e)|, DS:1017-1021) for each.
The keep rule of ``auto`` is derived from this table: keep the fast path's result wherever
e_tridiagonal <= e_ssyevd (x slack).  Runs on the GPU box; writes gpurun_out/r06_eigh_keep_rule.json.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import shampoo_oracle as orc
from precondition_amd import kernels as K

dev = torch.device("cuda:0")
F32 = np.float32


def haar(n, rng):
  q, r = np.linalg.qr(rng.standard_normal((n, n)))
  return q * np.sign(np.diag(r))


def make(kind, n, seed):
  rng = np.random.default_rng(seed)
  if kind.startswith("graded"):  # eigenvalues cond^(-i/(n-1))
    cond = float(kind[len("graded"):])
    q = haar(n, rng)
    a = (q * cond ** (-np.arange(n) / (n - 1))) @ q.T
  elif kind == "loguniform":  # 10^U(-4, 2)
    q = haar(n, rng)
    a = (q * 10.0 ** rng.uniform(-4, 2, n)) @ q.T
  elif kind == "lowrank":  # rank n/4 (+ the relative ridge of DS:1005)
    g = rng.standard_normal((n, n // 4))
    a = g @ g.T
  elif kind == "rank8_ema":  # statistics of rank-8 gradients, beta2 = 0.999, 40 updates from eps I
    a = 1e-6 * np.eye(n)
    for _ in range(40):
      g = rng.standard_normal((n, 8)) * 0.02
      a = 0.999 * a + 0.001 * (g @ g.T)
  elif kind == "wishart":
    g = rng.standard_normal((n, 2 * n))
    a = g @ g.T
  else:
    raise ValueError(kind)
  return ((a + a.T) / 2).astype(F32)


def gpu_root(a, p, solver):
  t = torch.tensor(a, device=dev)
  r, m = K.matrix_inverse_pth_root_batched([t], [p], [a.shape[0]], eigh=True,
                                           options={"eigh_solver": solver})
  torch.cuda.synchronize()
  return r[0].cpu().numpy().astype(np.float64), float(m[0, 0].item()), float(m[0, 7].item())


def main():
  sizes = [int(x) for x in os.environ.get("KEEP_SIZES", "169,512,1024,2048").split(",")]
  kinds = ["wishart", "graded1e2", "graded1e3", "graded1e4", "graded1e5", "graded1e6",
           "loguniform", "lowrank", "rank8_ema"]
  rows = []
  for n in sizes:
    for kind in kinds:
      for p in ((2, 4) if n <= 1024 else (2,)):
        a = make(kind, n, 1000 * p + n)
        t0 = time.time()
        truth = orc.eigh_root_float64(a, p)
        tn = np.linalg.norm(truth)
        row = dict(kind=kind, n=n, p=p)
        for nm, lp in (("ssyevd", "f32"), ("f64lapack", "f64")):
          h, m = orc.matrix_inverse_pth_root_eigh(a, p, lapack=lp)
          row["e_" + nm] = float(np.linalg.norm(h - truth) / tn)
          row["metric_" + nm] = float(m["inverse_pth_root_errors"])
        for solver in ("tridiagonal", "one_sided", "auto"):
          got, metric, cond = gpu_root(a, p, solver)
          row["e_" + solver] = float(np.linalg.norm(got - truth) / tn)
          row["metric_" + solver] = metric
          if solver == "tridiagonal":
            row["cond_reported"] = cond
        row["host_s"] = round(time.time() - t0, 1)
        rows.append(row)
        print(f"{kind:11s} n={n:5d} p={p} cond={row['cond_reported']:.2e}  ssyevd {row['e_ssyevd']:.2e} "
              f"| tridiagonal {row['e_tridiagonal']:.2e} one_sided {row['e_one_sided']:.2e} auto {row['e_auto']:.2e} "
              f"| metric ssyevd {row['metric_ssyevd']:.2e} td {row['metric_tridiagonal']:.2e} "
              f"os {row['metric_one_sided']:.2e}", flush=True)
  os.makedirs("gpurun_out", exist_ok=True)
  with open("gpurun_out/r06_eigh_keep_rule.json", "w") as f:
    json.dump(rows, f, indent=1)


if __name__ == "__main__":
  main()
