"""Dev: accuracy / cost of subspace.top_eigenpairs_batched vs numpy float64 eigh."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import subspace, kernels as K
dev = torch.device("cuda:0")

def check(name, mats, k, **kw):
  for _ in range(1): subspace.top_eigenpairs_batched(mats, k, **kw)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  e, v, conv, info = subspace.top_eigenpairs_batched(mats, k, **kw)
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
  a = mats[0].double().cpu().numpy()
  w = np.linalg.eigvalsh(a)[::-1][:k + 1]
  ee = e[0].double().cpu().numpy()
  vv = v[0].double().cpu().numpy()
  orth = np.abs(vv.T @ vv - np.eye(k)).max()
  resid = np.linalg.norm(a @ vv - vv * ee, axis=0).max() / w[0]
  print(f"{name}: {len(mats)} x {a.shape[0]}^2 k={k}: {dt*1e3:.1f} ms, converged {conv.tolist()[:3]}, {info}, "
        f"eig rel err {np.abs(ee - w[:k]).max() / w[0]:.2e}, orth {orth:.1e}, resid {resid:.1e}")

rng = np.random.default_rng(0)
def spectrum(n, vals):
  q, _ = np.linalg.qr(rng.standard_normal((n, n)))
  return torch.tensor(((q * vals) @ q.T).astype(np.float32), device=dev)
n = 1024
vals = np.concatenate([np.linspace(100, 20, 20), np.linspace(5, 0.1, n - 20)])
check("gapped", [spectrum(n, vals) for _ in range(2)], 17)
check("decay", [spectrum(n, 1000.0 * 0.97 ** np.arange(n))], 33)
g = torch.randn(1024, 30, device=dev); check("rank30", [g @ g.T], 65)
gs = [torch.randn(4096, 4096, device=dev) for _ in range(8)]
mats = [x @ x.T for x in gs]
check("gaussian4096", mats, 65)
check("gaussian4096 deg20", mats, 65, degree=20)
t0 = time.perf_counter(); es, us = K.eigh_batched(mats[:1]); torch.cuda.synchronize(); print("full eigh 1 x 4096^2: %.0f ms" % ((time.perf_counter() - t0) * 1e3))
