"""Dev (round 5): the optimizer on the HIP kernels (cuda) against the same host logic on tests/cpu_backend (the oracle's
arithmetic, CPU) on random trees and kwargs: updates of every step within a conditioning-sized tolerance."""
import os
os.environ.setdefault("PS_DEV_ENV", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import precondition_amd as pa
from tests import cpu_backend

dev = torch.device("cuda:0")
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 5e-3
bad = 0
worst = 0.0
for case in range(ncases):
  rng = np.random.default_rng(7000 * seed0 + case)
  shapes = []
  for _ in range(int(rng.integers(1, 6))):
    nd = int(rng.choice([0, 1, 1, 2, 2, 2, 3]))
    shapes.append(tuple(int(rng.choice([1, 3, 17, 64, 130, 200])) for _ in range(nd)))
  kw = dict(
      beta1=float(rng.choice([0.0, 0.9])), beta2=float(rng.choice([0.999, 1.0])),
      weight_decay=float(rng.choice([0.0, 1e-3])),
      start_preconditioning_step=int(rng.choice([1, 2])),
      preconditioning_compute_steps=int(rng.choice([1, 2, 3])),
      statistics_compute_steps=int(rng.choice([1, 2])),
      best_effort_shape_interpretation=bool(rng.integers(0, 2)),
      graft_type=pa.GraftingType(int(rng.integers(0, 7))),
      nesterov=bool(rng.integers(0, 2)),
      moving_average_for_momentum=bool(rng.integers(0, 2)),
      exponent_override=int(rng.choice([0, 0, 2])),
      precondtioner_type=pa.PreconditionerType(int(rng.choice([1, 2, 3]))),
      eigh=bool(rng.integers(0, 4) == 0),
      decoupled_learning_rate=bool(rng.integers(0, 2)), decoupled_weight_decay=bool(rng.integers(0, 2)),
      best_effort_memory_usage_reduction=bool(rng.integers(0, 5) == 0),
      clip_by_scaled_gradient_norm=None if rng.integers(0, 3) else 1.0,
      skip_preconditioning_rank_lt=int(rng.choice([1, 1, 2])),
      matrix_epsilon=float(rng.choice([1e-6, 1e-3])),
  )
  if os.environ.get("FUZZ_LOWRANK"):   # the compressed branches (DS:1033-1290)
    kw.update(eigh=False, compression_rank=int(rng.choice([2, 4, 8])), frequent_directions=bool(rng.integers(0, 2)),
              average_grad=bool(rng.integers(0, 2)), reset_preconditioner=bool(rng.integers(0, 2)),
              generate_fd_metrics=bool(rng.integers(0, 2)))
    if not kw["frequent_directions"]:
      kw.update(average_grad=False, reset_preconditioner=False, generate_fd_metrics=False)
    else:   # the reference's own constraints on the sketch branch
      kw.update(reuse_preconditioner=True, statistics_compute_steps=kw["preconditioning_compute_steps"])
  block = int(rng.choice([64, 128, 256]))
  p0 = [np.asarray(rng.standard_normal(s) * 0.1, np.float32) for s in shapes]
  def grads_at(t):
    r = np.random.default_rng(10_000 * case + t)
    return [np.asarray(r.standard_normal(s) * 0.1, np.float32) for s in shapes]
  outs = {}
  try:
    for where in ("cpu", "gpu"):
      d = torch.device("cpu") if where == "cpu" else dev
      opt = pa.distributed_shampoo(0.1, block, _backend_for_testing=cpu_backend if where == "cpu" else None, **kw)
      params = [torch.from_numpy(x).to(d) for x in p0]
      st = opt.init(params)
      ups = []
      for t in range(5):
        upd, st = opt.update([torch.from_numpy(g).to(d) for g in grads_at(t)], st, params)
        ups.append([u.detach().cpu().numpy().copy() for u in upd])
      outs[where] = ups
  except Exception as ex:
    import traceback
    tb = [l.strip() for l in traceback.format_exc().splitlines() if l.strip().startswith("File")][-1]
    print(f"case {case}: {where}: {type(ex).__name__}: {str(ex)[:160]}  at {tb}  shapes={shapes}", flush=True)
    bad += 1
    continue
  ok = True
  for t, (a, b) in enumerate(zip(outs["cpu"], outs["gpu"])):
    for i, (x, y) in enumerate(zip(a, b)):
      if not (np.isfinite(x).all() and np.isfinite(y).all()):
        if (np.isnan(x) != np.isnan(y)).any():
          ok = False; print(f"case {case}: NaN pattern differs step {t} leaf {i}", flush=True)
        continue
      rel = np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-30)
      worst = max(worst, rel)
      if rel > tol:
        ok = False
        print(f"case {case}: step {t} leaf {i} shape {x.shape} rel {rel:.2e}", flush=True)
  if not ok:
    bad += 1
    print(f"   kwargs { {k: (int(v) if hasattr(v, 'value') else v) for k, v in kw.items()} } block {block} shapes {shapes}", flush=True)
print(f"optimizer fuzz: {ncases} cases, {bad} flagged, worst rel {worst:.2e}")
