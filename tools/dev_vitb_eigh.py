"""Dev (rounds 5-6): the optimizer with eigh=True on a ViT-B/16-shaped tree (395 statistics of 197 ... 1024 rows, real
Shampoo statistics: rank-deficient early on): a few steps incl. recomputes, finite updates, time per recompute step, how
many blocks the fast path kept."""
import os
os.environ.setdefault("PS_DEV_ENV", "1")
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import precondition_amd as pa
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
L = 12
shapes = [(768, 768 * 3), (768 * 3,), (768, 768), (768,), (768, 3072), (3072,), (3072, 768), (768,), (768,), (768,)] * L
shapes += [(1, 197, 768), (768, 1000), (1000,), (16 * 16 * 3, 768)]
params = [torch.from_numpy(np.asarray(rng.standard_normal(s) * 0.02, np.float32)).to(dev) for s in shapes]
lowrank = len(sys.argv) > 1 and sys.argv[1] == "lowrank"   # gradients of rank 8: statistics stay ill conditioned
def make_grad(r, s):
  if lowrank and len(s) == 2 and min(s) > 8:
    # (plain loops, no BLAS: a threaded host matmul leaves its worker threads spinning into the timed region,
    # where they slow the enqueue of the ~25 000 launches of an eigh recompute -- round 5's 274-314 ms for this
    # mode were partly that)
    a, b = r.standard_normal((s[0], 8)), r.standard_normal((8, s[1]))
    return (np.einsum("ik,kj->ij", a, b, optimize=False) * 0.02 / 3).astype(np.float32)
  return np.asarray(r.standard_normal(s) * 0.02, np.float32)
# round 6: "auto" keeps every block's fast-path result (at or below a true float32 ssyevd's root error);
# "accurate" is round 5's default (Jacobi hand-over above cond 1e3, with / without the optimizer's memo)
configs = (("auto", True), ("accurate", True), ("accurate", False), ("one_sided", True))
if os.environ.get("VITB_ONLY"):
  configs = tuple(c for c in configs if c[0] == os.environ["VITB_ONLY"])[:1]
for solver, hint in configs:
  opt = pa.distributed_shampoo(0.1, 1024, preconditioning_compute_steps=2, start_preconditioning_step=1, eigh=True,
                               eigh_solver=solver, graft_type=pa.GraftingType.RMSPROP_NORMALIZED,
                               iteration_count_hint=hint)
  st = opt.init(params)
  times = []
  for t in range(14):
    r = np.random.default_rng(100 + t)
    grads = [torch.from_numpy(make_grad(r, s)).to(dev) for s in shapes]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    upd, st = opt.update(grads, st, params)
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    ok = all(bool(torch.isfinite(u).all()) for u in upd)
    if not ok: print(f"{solver} hint={hint}: step {t}: non-finite update", flush=True)
  rec = times[0::2]
  print(f"{solver} hint={hint}: recompute steps (ms): " + " ".join(f"{x:.0f}" for x in rec) + f"   median of the last five {sorted(rec[-5:])[2]:.0f}", flush=True)
