"""Dev: run the optimizer over a matrix of options / shapes for a few steps and check that
nothing crashes and everything stays finite."""
import itertools, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import precondition_amd as pa
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
shape_sets = {
    "mixed": ([300, 200], [7], [3, 5, 7, 11], [1, 1, 64], [130, 1], [2, 3, 4, 5, 6], [513]),
    "wide": ([1000, 40], [40, 1000], [129, 129]),
}
def mk(shapes):
  return tuple(torch.tensor(rng.standard_normal(s).astype(np.float32), device=dev) for s in shapes)
opts = []
for gt in pa.GraftingType:
  opts.append(dict(graft_type=gt))
for pt in pa.PreconditionerType:
  opts.append(dict(precondtioner_type=pt, merge_small_dims_block_size=1) if pt != pa.PreconditionerType.ALL else dict(precondtioner_type=pt))
opts += [dict(nesterov=False, moving_average_for_momentum=True, weight_decay=0.01),
         dict(decoupled_weight_decay=True, decoupled_learning_rate=False, weight_decay=0.1),
         dict(beta2=1.0), dict(exponent_override=3), dict(eigh=True),
         dict(best_effort_memory_usage_reduction=True),
         dict(compression_rank=3), dict(compression_rank=-2),
         dict(compression_rank=4, frequent_directions=True, reuse_preconditioner=True, statistics_compute_steps=2, preconditioning_compute_steps=2),
         dict(lobpcg_topk_precondition=2), dict(clip_by_scaled_gradient_norm=0.5, graft_type=pa.GraftingType.RMSPROP),
         dict(skip_preconditioning_rank_lt=2), dict(skip_preconditioning_dim_size_gt=400),
         dict(best_effort_shape_interpretation=False), dict(generate_training_metrics=False),
         dict(learning_rate_fn=True)]
fails = 0
for (sname, shapes), kw in itertools.product(shape_sets.items(), opts):
  kw = dict(kw)
  for bs in (64, 128):
    try:
      lr = (lambda t: 0.1 / (1 + t)) if kw.pop("learning_rate_fn", False) else 0.1
      base = dict(preconditioning_compute_steps=2, start_preconditioning_step=1)
      base.update(kw)
      if (kw.get("precondtioner_type") in (pa.PreconditionerType.INPUT, pa.PreconditionerType.OUTPUT)):
        sh = [s for s in shapes if len(s) >= 2 and min(s) > 1]
      else:
        sh = shapes
      params = mk(sh)
      opt = pa.distributed_shampoo(lr, bs, **base)
      st = opt.init(params)
      for t in range(4):
        g = mk(sh)
        upd, st = opt.update(g, st, params)
        for u, p in zip(upd, params):
          assert u.shape == p.shape and torch.isfinite(u).all(), "non-finite update"
    except Exception as e:  # noqa
      fails += 1
      print("FAIL", sname, bs, kw, "->", type(e).__name__, str(e)[:200])
      traceback.print_exc(limit=3)
print("done, failures:", fails, "of", 2 * len(opts) * len(shape_sets))
