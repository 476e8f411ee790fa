"""Dev (round 5): distributed_shampoo(donate_state=True) against the functional path on random trees and kwargs:
updates and final state must agree bit for bit over several steps incl. recomputes (or the factory must refuse
the combination loudly)."""
import os
os.environ.setdefault("PS_DEV_ENV", "1")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import precondition_amd as pa

dev = torch.device("cuda:0")
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for case in range(ncases):
  rng = np.random.default_rng(1000 * seed0 + case)
  shapes = []
  for _ in range(int(rng.integers(1, 7))):
    nd = int(rng.choice([0, 1, 1, 2, 2, 2, 3]))
    shapes.append(tuple(int(rng.choice([1, 3, 17, 64, 130, 200, 257])) for _ in range(nd)))
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
      precondtioner_type=pa.PreconditionerType(int(rng.choice([1, 2, 3]))) if hasattr(pa, "PreconditionerType") else 1,
      eigh=bool(rng.integers(0, 4) == 0),
      decoupled_learning_rate=bool(rng.integers(0, 2)), decoupled_weight_decay=bool(rng.integers(0, 2)),
      best_effort_memory_usage_reduction=bool(rng.integers(0, 5) == 0),
      clip_by_scaled_gradient_norm=None if rng.integers(0, 3) else 1.0,
      skip_preconditioning_rank_lt=int(rng.choice([1, 1, 2])),
  )
  block = int(rng.choice([64, 128, 256]))
  params = [torch.from_numpy(np.asarray(rng.standard_normal(s) * 0.1, np.float32)).to(dev) for s in shapes]
  def grads_at(t):
    r = np.random.default_rng(10_000 * case + t)
    return [torch.from_numpy(np.asarray(r.standard_normal(s) * 0.1, np.float32)).to(dev) for s in shapes]
  outs = {}
  try:
    for donate in (False, True):
      opt = pa.distributed_shampoo(0.1, block, donate_state=donate, **kw)
      st = opt.init(params)
      ups = []
      for t in range(6):
        upd, st = opt.update(grads_at(t), st, params)
        ups.append([u.clone() for u in upd])
      outs[donate] = (ups, [x.clone() if isinstance(x, torch.Tensor) else x for x in pa.pytree.tree_leaves(st)])
  except Exception as ex:
    print(f"case {case}: {type(ex).__name__}: {str(ex)[:150]}  shapes={shapes} donate={donate}", flush=True)
    continue
  ok = True
  for t, (a, b) in enumerate(zip(outs[False][0], outs[True][0])):
    for i, (x, y) in enumerate(zip(a, b)):
      if not torch.equal(x, y) and not (torch.isnan(x) & torch.isnan(y)).all():
        ok = False
        print(f"case {case}: UPDATE differs at step {t} leaf {i} shape {tuple(x.shape)} max {float((x - y).abs().max()):.3e}", flush=True)
        break
    if not ok: break
  if ok:
    for i, (x, y) in enumerate(zip(outs[False][1], outs[True][1])):
      if isinstance(x, torch.Tensor) and not torch.equal(x, y) and not (x != x).any():
        ok = False
        print(f"case {case}: STATE leaf {i} differs shape {tuple(x.shape)}", flush=True)
        break
  if not ok:
    bad += 1
    print(f"   kwargs {kw} block {block} shapes {shapes}", flush=True)
print(f"donate fuzz: {ncases} cases, {bad} mismatches")
