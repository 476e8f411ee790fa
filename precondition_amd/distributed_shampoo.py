"""Distributed Shampoo optimizer surface on the MI355X preconditioner kernels.

Same factory signature, state types, step semantics and construction-time
errors as the reference's ``distributed_shampoo`` (precondition/
distributed_shampoo.py, DS:1849-2040 factory, DS:2585-2625 init_fn, DS:2631-2675
_compute_stats, DS:2816-3010 _pmap_compute_preconditioners, DS:3442-3494
_compute_preconditioners, DS:3496-3625 _transform_grad, DS:3627-3659 update_fn),
re-hosted on torch tensors:

  * statistics accumulation of a whole parameter tree = one grouped launch of the
    HIP Gram kernel (kernels.stats_update_grouped);
  * the preconditioner recompute = one batched call of the HIP Newton / eigh root
    over the statistics this rank owns, then one RCCL all-gather of the roots and
    one of the metrics table (comm.py) instead of jax.lax.all_gather;
  * grafting / momentum (_transform_grad) are elementwise torch ops (row f2 of
    SURVEY.md §8, "next").

``batch_axis_name`` keeps its role as the switch for data-parallel sharding of
the statistics blocks: pass a ``torch.distributed`` process group (or any truthy
value for the default group).  There is no CPU fallback: the statistics and root
steps raise without libprecondition_amd.so and an MI355X.
"""
from __future__ import annotations

import contextlib
import gc
import logging
import os
from typing import Any, Callable, List, Optional

import numpy as np
import torch

from . import comm
from . import pytree
from .blocking import Preconditioner, _precond_dim, _should_compress
from .state import (_FD_FIELDS, FDDiagnostics, GlobalShardedParameterStats,
                    GradientTransformation, GraftingType, InitFnState,
                    LocalShardedParameterStats, MaskedNode, ParameterStats, PreconditionerType,
                    QuantizedValue, ShampooState, ShardedShampooStats, TrainingMetrics,
                    init_training_metrics)

_EPSILON = 1e-25  # DS:41


def _pth_root_difference(w, alpha, beta, p):
  """(w+alpha)^(-1/p) - (w+beta)^(-1/p) without cancellation (DS:681-699); float32
  scalars or tensors, evaluated where the arguments live (host math, O(rank) values:
  the LOBPCG re-deflation weights)."""
  w, alpha, beta = (torch.as_tensor(x, dtype=torch.float32) for x in (w, alpha, beta))
  a = w + alpha
  b = w + beta
  a_minus_b = alpha - beta
  exp = -1.0 / float(p)

  def _stable_subtract(base, diff):
    # (base)^exp * expm1(exp * log1p(diff / base)): the common factor pulled out
    return torch.pow(base, exp) * torch.expm1(exp * torch.log1p(diff / base))

  return torch.where(torch.abs(a_minus_b / b) < torch.abs(a_minus_b / a),
                     -_stable_subtract(a, -a_minus_b), _stable_subtract(b, a_minus_b))


def preconditioning_compute_steps_schedule(lr_fn, start_preconditioning_compute_steps,
                                           end_preconditioning_compute_steps, step):
  """DS:44-76: recompute interval following the learning-rate decay, rounded
  down to a multiple of 10, at least 1."""
  decay_factor = float(lr_fn(step)) / float(lr_fn(0))
  t = (start_preconditioning_compute_steps +
       (1 - decay_factor) * end_preconditioning_compute_steps)
  return max((t // 10) * 10, 1)


@contextlib.contextmanager
def _collector_paused():
  """update() of a few-hundred-leaf tree allocates ~10^4 short-lived containers (block views,
  descriptor rows, state tuples) and creates NO reference cycles (tools/dev_find_cycles.py).
  At CPython's thresholds that churn alone triggers ~14 young, ~1.4 middle and ~0.14 FULL
  collections per call; the full ones walk every object of the process (torch, numpy, the
  optimizer state) and cost ~3 ms per update() on the ViT-B tree, amortised.  The collector is
  therefore paused for the duration of the call and restored on exit: what it would have found
  is still found, by one young-generation pass after the call.  PS_UPDATE_KEEP_GC=1 opts out."""
  paused = gc.isenabled() and os.environ.get("PS_UPDATE_KEEP_GC", "0") != "1"
  if paused:
    gc.disable()
  try:
    yield
  finally:
    if paused:
      gc.enable()


def distributed_shampoo(
    learning_rate,
    block_size,
    beta1=0.9,
    beta2=0.999,
    diagonal_epsilon=1e-10,
    matrix_epsilon=1e-6,
    weight_decay=0.0,
    start_preconditioning_step=5,
    preconditioning_compute_steps=1,
    decay_preconditioning_compute_steps: bool = False,
    end_preconditioning_compute_steps: Optional[int] = None,
    statistics_compute_steps=1,
    best_effort_shape_interpretation=True,
    graft_type=GraftingType.SGD,
    nesterov=True,
    exponent_override=0,
    batch_axis_name=None,
    statistics_partition_spec=None,
    preconditioner_partition_spec=None,
    num_devices_for_pjit=None,
    shard_optimizer_states=False,
    best_effort_memory_usage_reduction=False,
    inverse_failure_threshold=0.1,
    moving_average_for_momentum=False,
    skip_preconditioning_dim_size_gt=4096,
    clip_by_scaled_gradient_norm=None,
    precision=None,
    tensordot_precision=None,
    relative_matrix_epsilon=True,
    merge_small_dims_block_size=4096,
    lobpcg_topk_precondition: int = 0,
    lobpcg_max_iter: int = 0,
    precondtioner_type=PreconditionerType.ALL,  # (sic) the reference's spelling
    generate_fd_metrics: bool = False,
    compression_rank: int = 0,
    frequent_directions: bool = False,
    reset_preconditioner: bool = False,
    average_grad: bool = False,
    skip_preconditioning_rank_lt=1,
    decoupled_learning_rate=True,
    decoupled_weight_decay=False,
    generate_training_metrics=True,
    reuse_preconditioner=False,
    eigh=False,
    # build-specific (not in the reference): ownership of statistics across
    # ranks ("reference" = batch() order of DS:1827, "lpt" = cost balanced), and
    # a test seam (tests/ inject a CPU stand-in to exercise the host logic and
    # the gloo sharding without a GPU; None = the HIP kernels, which fail loudly
    # when the library or the GPU is missing).
    block_ownership: str = "reference",
    # Owner-only statistics: with a process group, a rank keeps and updates
    # only the statistics it roots; the others are empty placeholders in its state.
    # Memory and Gram FLOPs of the statistics drop by the group size; preconditioners
    # stay replicated (every rank applies them to its full gradient).  The optimizer
    # state is then rank-local: checkpoint every rank (or gather) to save it.
    shard_statistics: bool = False,
    # eigh=True only: which eigensolver roots the blocks of more than 128 rows (ps_options.eigh_solver).
    # "auto" = tridiagonalisation + divide and conquer (the class of the reference's float32 LAPACK
    # ssyevd, DS:35-38 / DS:1007), result kept for every block: at or below ssyevd's own root error on
    # every statistics family measured (profiles/r06_eigh_keep_rule.json).  "accurate" = blocks whose
    # spectrum spans more than 1e3 -- rank-deficient statistics + ridge among them -- are solved again
    # by the one-sided Jacobi solver inside the call (10-300 x closer to the float64 root than the
    # reference itself on such blocks, 2-3 x the time; the default until round 6); "two_sided"
    # reproduces a float64-internal LAPACK result on such blocks to three digits (include/ps_api.h).
    eigh_solver: str = "auto",
    # Newton branch: the previous recompute's iteration counts (state.training_metrics) steer the
    # per-block accuracy policy of the next root call (ps_options.iters_hint: blocks that took <= 8
    # iterations skip the averaged M updates) and weight the "lpt" ownership.  An explicit option
    # of this build (it needs generate_training_metrics=True to have the counts in the state, but is
    # not implied by it): False = every block takes the careful path on every recompute.
    iteration_count_hint: bool = True,
    # The counterpart of the reference's jit-compiled update_fn (DS:3627-3659: traced once, no
    # per-step host work): update() takes ownership of the state it is given.  On the steps without
    # a root recompute the statistics, diagonal statistics and momenta are updated IN PLACE, the
    # returned state holds the same tensors, and the returned updates live in buffers the optimizer
    # owns (valid until the next update call) -- no allocation, no descriptor rebuilt, the same
    # seven launches (plan.DonatedStep).  The state passed in must not be used afterwards.  Applies
    # to float32 dense states (the tree-plan path); anything else takes the functional path.
    donate_state: bool = False,
    _backend_for_testing: Any = None,
):
  """Returns GradientTransformation(init_fn, update_fn); see module docstring."""
  # `precision` (DS:599, 708, 1883: the jax.lax.Precision of the Newton products) selects the
  # product arithmetic of the root calls: HIGHEST / None -> exact float32 MFMA (the parity
  # path, the reference's default), HIGH -> bf16x6, DEFAULT -> bf16x3 (kernels.products_for_
  # precision, ps_options.products).  `tensordot_precision` (the Gram updates, DS:1469) is
  # validated the same way; the statistics kernel always accumulates exact float32 products
  # (None, the reference's default there, means "highest the backend has").
  from .kernels import products_for_precision
  root_options = {"products": products_for_precision(precision)}
  if eigh_solver not in ("auto", "accurate", "tridiagonal", "one_sided", "two_sided"):
    raise ValueError("eigh_solver must be auto | accurate | tridiagonal | one_sided | two_sided, "
                     f"found {eigh_solver!r}")
  if eigh and eigh_solver != "auto":
    root_options["eigh_solver"] = eigh_solver
  _eigh_memo = {"cond": None}   # _compute_preconditioners: every block's condition number at the last eigh recompute
  products_for_precision(tensordot_precision)
  del tensordot_precision
  # (the partition specs describe XLA shardings; here the stacked statistics are always
  # split along their leading axis over the ranks of the process group)
  del statistics_partition_spec, preconditioner_partition_spec

  # ---- construction-time validation: same conditions and messages as DS:2019-2040
  if reset_preconditioner and not frequent_directions:
    raise ValueError("reset_preconditioner=True requries frequent_directions")
  generate_fd_metrics = generate_fd_metrics and frequent_directions
  if frequent_directions and compression_rank <= 0:
    raise ValueError("frequent_directions=True requires compression_rank > 0,"
                     f" found {compression_rank}")
  if average_grad and not frequent_directions:
    raise ValueError("average_grad requested but frequent_directions is False")
  if frequent_directions and (statistics_compute_steps !=
                              preconditioning_compute_steps):
    raise ValueError("frequent_directions=True requires "
                     f"statistics_compute_steps ({statistics_compute_steps}) "
                     "to equal != preconditioning_compute_steps "
                     f"({preconditioning_compute_steps})")
  # ---- scope of this build
  if shard_optimizer_states and block_ownership != "reference":
    raise ValueError("shard_optimizer_states keeps the reference's batch() ownership")
  if lobpcg_topk_precondition and (eigh or compression_rank != 0):
    raise NotImplementedError("lobpcg_topk_precondition is built for the dense Newton branch")
  reset_frequency = None
  if reset_preconditioner:  # DS:2022-2024
    reset_frequency = int(np.round(1 / (1 - beta2))) if beta2 != 1 else None
    beta2 = 1.0

  group = comm.resolve_group(batch_axis_name)
  # sharded optimizer state (DS:2162-2583) = owner-only statistics behind the stacked-array API
  shard_stats = bool((shard_statistics or shard_optimizer_states) and group is not None)
  if shard_optimizer_states and num_devices_for_pjit is not None:
    if int(num_devices_for_pjit) != comm.world_and_rank(group)[0]:
      raise ValueError(
          f"num_devices_for_pjit={num_devices_for_pjit} but the process group has "
          f"{comm.world_and_rank(group)[0]} rank(s): one process per GPU")
  if shard_stats and (frequent_directions or compression_rank != 0):
    raise NotImplementedError("shard_statistics supports the dense preconditioner mode")

  def _zeros_f32_like(param):
    """State buffers are float32 whatever the parameter dtype (the kernels take float32 only;
    bf16 / fp16 parameters and gradients are promoted at the door of update())."""
    return torch.zeros_like(param, dtype=torch.float32)

  def _owned_mask(params_flat):
    """mask[p][j]: does this rank own statistic j of parameter p (list order = the
    order _compute_preconditioners flattens them in)?  Pure function of shapes."""
    sizes, exps, counts = [], [], []
    for param in params_flat:
      n = 0
      if not _skip_preconditioning(param):
        pc = preconditioner_from_params(param)
        e = pc.exponent_for_preconditioner() if exponent_override == 0 else exponent_override
        for sh in pc.shapes_for_preconditioners():
          sizes.append(int(sh[0])); exps.append(e); n += 1
      counts.append(n)
    world, rank = comm.world_and_rank(group)
    owner = comm.ownership_table(sizes, exps, world, block_ownership)
    mask, k = [], 0
    for n in counts:
      mask.append([owner[k + j] == rank for j in range(n)])
      k += n
    return mask
  if _backend_for_testing is not None:
    backend = _backend_for_testing
  else:
    from . import kernels as backend  # HIP path

  def _graft_type_has_diagonal_statistics():
    return graft_type not in (GraftingType.SGD, GraftingType.SQRT_N, GraftingType.NONE)

  def _quantize(x):
    return QuantizedValue.from_float_value(x, torch.float32)

  # ---- quantized optimizer state (best_effort_memory_usage_reduction) -------------
  # DS:2047-2070: momentum buffers of rank > 1 parameters are int8; statistics and
  # preconditioners are int16 with the diagonal kept in float32, but only in the
  # sharded (batch_axis_name) dense mode.
  def quantized_dtype_for_momentum_buffers(var):
    return torch.int8 if (best_effort_memory_usage_reduction and
                          len(var.shape) > 1) else torch.float32

  # pjit mode (shard_optimizer_states): the stacked statistics / preconditioners are float32
  # arrays (DS:2250-2252, quantized=False at DS:2557); only the momentum buffers are int8
  quantize_second_moment = bool(best_effort_memory_usage_reduction and
                                not compression_rank and not frequent_directions and
                                batch_axis_name and not shard_optimizer_states)
  qdt_second_moment = torch.int16 if quantize_second_moment else torch.float32

  def _quantize_many(tensors, dtype, extract_diagonal):
    """QuantizedValue.from_float_value for a list of tensors in one grouped launch."""
    if dtype == torch.float32:
      return [QuantizedValue.from_float_value(t, torch.float32, extract_diagonal)
              for t in tensors]
    full = [i for i, t in enumerate(tensors) if t.numel() > 0]
    triples = backend.quantize_grouped([tensors[i] for i in full], dtype, extract_diagonal)
    out = [None] * len(tensors)
    for i, (q, d, b) in zip(full, triples):
      out[i] = QuantizedValue(q, d, b, dtype, extract_diagonal, list(q.shape))
    for i, t in enumerate(tensors):
      if out[i] is None:  # empty placeholder (a statistic another rank owns)
        zf = torch.empty((0,), dtype=torch.float32, device=t.device)
        out[i] = QuantizedValue(torch.empty(t.shape, dtype=dtype, device=t.device), zf, zf,
                                dtype, extract_diagonal, list(t.shape))
    return out

  def _to_float_many(values):
    """_to_float (DS:2072-2076) for a list of QuantizedValue / tensors, grouped."""
    out = list(values)
    idx = [i for i, v in enumerate(values) if isinstance(v, QuantizedValue) and
           v.quantized_dtype in (torch.int8, torch.int16)]
    for i in idx:
      if values[i].quantized.numel() == 0:  # empty placeholder
        out[i] = torch.empty(tuple(values[i].quantized.shape), dtype=torch.float32,
                             device=values[i].quantized.device)
    idx = [i for i in idx if values[i].quantized.numel() > 0]
    for extract in (False, True):  # one grouped launch per kind
      sel = [i for i in idx if bool(values[i].extract_diagonal) == extract]
      if not sel:
        continue
      fl = backend.dequantize_grouped([
          (values[i].quantized, values[i].diagonal if extract else [], values[i].bucket_size)
          for i in sel])
      for i, f in zip(sel, fl):
        out[i] = f
    for i, v in enumerate(out):
      if isinstance(v, QuantizedValue):
        out[i] = v.to_float()  # float32 / bfloat16 pass-through
    return out

  def _quantize_momentum_many(moms, params):
    """_quantize_momentum (DS:2111-2114) for the whole tree."""
    out = [None] * len(moms)
    q_idx = [i for i, p in enumerate(params)
             if quantized_dtype_for_momentum_buffers(p) == torch.int8]
    if q_idx:
      for i, qv in zip(q_idx, _quantize_many([moms[i] for i in q_idx], torch.int8, False)):
        out[i] = qv
    for i, m in enumerate(moms):
      if out[i] is None:
        out[i] = _quantize(m)
    return out

  _pc_cache = {}

  def preconditioner_from_params(param):
    """Bookkeeping depends on the shape only: one Preconditioner per distinct shape."""
    key = tuple(param.shape)
    pc = _pc_cache.get(key)
    if pc is None:
      pc = Preconditioner(param, block_size, merge_small_dims_block_size,
                          best_effort_shape_interpretation, precondtioner_type,
                          compression_rank)
      _pc_cache[key] = pc
    return pc

  def _skip_preconditioning(param):
    return len(param.shape) < skip_preconditioning_rank_lt or any(
        s > skip_preconditioning_dim_size_gt for s in param.shape)

  _plan_cache = {}

  def _tree_plan(params_flat):
    """plan.TreePlan of this tree (shapes only), or None when the tree or the options need
    the general per-block path: FD / compression, owner-only statistics, blocks of merged
    rank > 2, the host-logic test backend."""
    if (not hasattr(backend, "transform_grads_fused") or frequent_directions or
        compression_rank or shard_stats or
        os.environ.get("PS_UPDATE_PLAN", "1") == "0"):
      return None
    key = tuple(tuple(p.shape) for p in params_flat)
    if key not in _plan_cache:
      from .plan import TreePlan
      skipped = [_skip_preconditioning(p) for p in params_flat]
      pcs = [None if sk else preconditioner_from_params(p)
             for p, sk in zip(params_flat, skipped)]
      _plan_cache[key] = TreePlan.build(key, pcs, skipped)
    return _plan_cache[key]

  def _placeholder_like(x):
    """A statistic this rank does not own: no storage."""
    if isinstance(x, QuantizedValue):
      z = torch.empty((0, 0), dtype=x.quantized.dtype, device=x.quantized.device)
      zf = torch.empty((0,), dtype=torch.float32, device=x.quantized.device)
      return QuantizedValue(z, zf, zf, x.quantized_dtype, x.extract_diagonal, [0, 0])
    return torch.empty((0, 0), dtype=x.dtype, device=x.device)

  # ---------------------------------------------------------------------------
  def init_fn(params):
    """DS:2585-2625: statistics = matrix_epsilon * I, preconditioners = I."""

    def _init(param):
      dev = param.device
      statistics, preconditioners = [], []
      if not _skip_preconditioning(param):
        shapes = preconditioner_from_params(param).shapes_for_preconditioners()
        statistics = [
            matrix_epsilon * torch.eye(s[0], dtype=torch.float32, device=dev)
            for s in shapes
        ]
        preconditioners = [
            torch.eye(s[0], s[1], dtype=torch.float32, device=dev) * float(s[0] == s[1])
            for s in shapes
        ]
      diagonal_statistics = []
      if _graft_type_has_diagonal_statistics():
        diagonal_statistics = _zeros_f32_like(param)
      if quantize_second_moment:  # DS:2613-2614
        statistics = _quantize_many(statistics, qdt_second_moment, True)
        preconditioners = _quantize_many(preconditioners, qdt_second_moment, True)
      zeros_q = _quantize_momentum_many([_zeros_f32_like(param), _zeros_f32_like(param)],
                                        [param, param])  # DS:2608-2609
      return ParameterStats(
          _quantize(diagonal_statistics), statistics, preconditioners,
          zeros_q[0], zeros_q[1],
          _zeros_f32_like(param) if (frequent_directions and average_grad) else MaskedNode(),
          init_training_metrics(len(statistics), generate_training_metrics,
                                generate_fd_metrics, device=dev))

    stats = pytree.tree_map(_init, params)
    if frequent_directions and _backend_for_testing is None:
      # init-time preparation of the sketch-update path (kernel loading + allocator pool) for
      # every (dimension, count) of compressed factors on the GPU: low_rank.prepare_fd
      from . import low_rank as _lr
      census = {}
      _, treedef_ = pytree.tree_flatten(params)
      for st_ in treedef_.flatten_up_to(stats):
        if isinstance(st_, ParameterStats):
          for pc_ in st_.preconditioners:
            if (isinstance(pc_, torch.Tensor) and pc_.is_cuda and pc_.dim() == 2 and
                pc_.shape[1] < pc_.shape[0]):
              key = (int(pc_.shape[0]), pc_.device)
              census[key] = census.get(key, 0) + 1
      for (d_, dev_), cnt in census.items():
        _lr.prepare_fd(d_, compression_rank, cnt, dev_)
    if shard_stats:
      params_flat, treedef = pytree.tree_flatten(params)
      st_flat = treedef.flatten_up_to(stats)
      mask = _owned_mask(params_flat)
      st_flat = [
          st._replace(statistics=[x if own else _placeholder_like(x)
                                  for x, own in zip(st.statistics, m)])
          for st, m in zip(st_flat, mask)]
      stats = treedef.unflatten(st_flat)
    return ShampooState(count=torch.zeros([], dtype=torch.int32), stats=stats)

  # ---------------------------------------------------------------------------
  def _compute_stats_all(grads_flat, stats_flat, params_flat, step):
    """DS:2631-2675 for every parameter, as ONE grouped Gram launch."""
    w1 = beta2
    w2 = beta2 if beta2 == 1.0 else 1.0 - beta2  # DS:2635-2636
    perform = statistics_compute_steps <= 1 or step % statistics_compute_steps == 0
    plan = _tree_plan(params_flat) if perform else None
    if plan is not None and all(g.is_contiguous() for g in grads_flat):
      # every (block, axis) addressed as `gradient pointer + planned offset`: no block views
      olds = [s for st in stats_flat for s in st.statistics]
      if quantize_second_moment and len(olds) == len(plan.stat_dims):
        # int16 state (DS:2652-2654): dequantize the tree in one launch, update the float
        # temporaries in place, quantize them back
        fl = _to_float_many(olds)
        if all(s.is_contiguous() for s in fl):
          plan.stats_update(grads_flat, fl, fl, w1, w2)
          qs = _quantize_many(fl, qdt_second_moment, True)
          out, k = [], 0
          for st, cnt in zip(stats_flat, plan.n_stats_of):
            out.append(ParameterStats(st.diagonal_statistics, qs[k:k + cnt], st.preconditioners,
                                      st.diagonal_momentum, st.momentum, MaskedNode(),
                                      st.training_metrics))
            k += cnt
          return out
      elif len(olds) == len(plan.stat_dims) and all(s.is_contiguous() for s in olds):
        news = [torch.empty_like(s) for s in olds]
        plan.stats_update(grads_flat, olds, news, w1, w2)
        out, k = [], 0
        for st, cnt in zip(stats_flat, plan.n_stats_of):
          out.append(ParameterStats(st.diagonal_statistics, news[k:k + cnt], st.preconditioners,
                                    st.diagonal_momentum, st.momentum, MaskedNode(),
                                    st.training_metrics))
          k += cnt
        return out
    new_lists, items, fd_items, new_avg = [], [], [], []
    owned = _owned_mask(params_flat) if shard_stats else None
    float_old = None
    if quantize_second_moment and perform:  # to_float=_to_float, DS:2652
      flat = _to_float_many([q for st in stats_flat for q in st.statistics])
      float_old, k = [], 0
      for st in stats_flat:
        float_old.append(flat[k:k + len(st.statistics)])
        k += len(st.statistics)
    for pidx, (grad, state, param) in enumerate(zip(grads_flat, stats_flat, params_flat)):
      avg = MaskedNode()
      if _skip_preconditioning(param):
        new_lists.append([[]] * len(state.statistics))
        new_avg.append(avg)
        continue
      if frequent_directions and average_grad:  # DS:2640-2645
        if statistics_compute_steps == 1 or step % statistics_compute_steps == 1:
          avg = grad
        else:
          avg = state.avg_grad + grad
        grad = avg / statistics_compute_steps
      new_avg.append(avg)
      if not perform:
        new_lists.append(state.statistics)
        continue
      pc = preconditioner_from_params(param)
      olds = [s.contiguous() for s in (float_old[pidx] if float_old is not None
                                       else state.statistics)]
      # quantized mode: the dequantized copies are temporaries, update them in place
      news = olds if float_old is not None else [torch.empty_like(s) for s in olds]
      for j, it in enumerate(pc.statistics_update_items(olds, grad, news)):
        if owned is not None and not owned[pidx][j]:
          news[j] = olds[j]  # not ours: stays an empty placeholder
          continue
        g_blk, axis = it[0], it[1]
        if frequent_directions and _should_compress(compression_rank, g_blk.shape[axis]):
          # FD (DS:1585-1588): the slot holds the Gram matrix of the (averaged)
          # gradient block itself (w1/w2 ignored, DS:1496).  The reference stores a
          # triangular factor R of it; only R R^T is ever consumed (DS:1179-1193).
          fd_items.append(it)
        else:
          items.append(it)
      new_lists.append(news)
    if items:
      backend.stats_update_grouped(items, w1, w2)
    if fd_items:
      backend.stats_update_grouped(fd_items, 0.0, 1.0)
    if float_old is not None:  # from_float=_maybe_quantize_statistics, DS:2654
      upd = [i for i, (st, p) in enumerate(zip(stats_flat, params_flat))
             if not _skip_preconditioning(p) and len(st.statistics)]
      qs = _quantize_many([t for i in upd for t in new_lists[i]], qdt_second_moment, True)
      k = 0
      for i in upd:
        new_lists[i] = qs[k:k + len(new_lists[i])]
        k += len(new_lists[i])
    return [
        ParameterStats(s.diagonal_statistics, ns, s.preconditioners,
                       s.diagonal_momentum, s.momentum, av, s.training_metrics)
        for s, ns, av in zip(stats_flat, new_lists, new_avg)
    ]

  # ---------------------------------------------------------------------------
  def _compute_preconditioners(states, params, step):
    """DS:3442-3494 + DS:2816-3010: shard the statistics over the ranks of
    `group`, root them, all-gather, select against the previous value."""
    statistics, exponents, prev, counts, klens = [], [], [], [], []
    for state, param in zip(states, params):
      num = len(state.statistics)
      counts.append(num)
      if num > 0:
        pc = preconditioner_from_params(param)
        e = pc.exponent_for_preconditioner() if exponent_override == 0 else exponent_override
        exponents.extend([e] * num)
        statistics.extend(state.statistics)
        prev.extend(state.preconditioners)
        klens.extend(pc.contraction_lengths())
    if not statistics:
      return states

    steps_t = preconditioning_compute_steps
    if (decay_preconditioning_compute_steps and end_preconditioning_compute_steps
        and callable(learning_rate)):
      steps_t = preconditioning_compute_steps_schedule(
          learning_rate, preconditioning_compute_steps,
          end_preconditioning_compute_steps, step)
    perform_step = step % steps_t == 0
    if not perform_step:
      # DS:2911-2926: "error = threshold" sentinels make the select keep every
      # previous preconditioner; metrics keep their old values (DS:2983-2986).
      return states

    if quantize_second_moment:
      # _quantized_matrix_inverse_pth_root_vmap (DS:2746-2773): the root is taken of the
      # DEQUANTIZED statistics and quantized again before the failure select.
      statistics = _to_float_many(statistics)
    # sizes from the (replicated) preconditioners: with shard_statistics the statistics
    # this rank does not own are empty placeholders
    sizes = [int(p.shape[0]) for p in prev]
    compute_fn, out_cols = None, None
    fd_cols = len(_FD_FIELDS) if (generate_fd_metrics and generate_training_metrics) else 0
    metrics_cols = comm.METRICS_STRIDE + fd_cols
    if compression_rank != 0:
      max_size = max(sizes)
      assert _precond_dim(compression_rank, max_size) < max_size, (
          "all layers are too small for compression_rank")  # DS:2127-2133
      out_cols = [_precond_dim(compression_rank, n) for n in sizes]
      if frequent_directions:
        assert reuse_preconditioner, "frequent_directions needs the previous sketch (DS:1137)"

      def prev_for(i):  # DS:2140-2154 (reset) ; true-size, so no padding needed
        pp = prev[i]
        if reset_frequency is not None and step % reset_frequency == 0:
          pp = torch.zeros_like(pp)
        return pp

      def compute_fn(indices, outs):  # new_mi_pth_root dispatch, DS:2706-2738
        rows = torch.zeros((len(indices), metrics_cols), dtype=torch.float32,
                           device=statistics[0].device)
        dense = [k for k, i in enumerate(indices)
                 if not _should_compress(compression_rank, sizes[i])]
        if dense:
          _, m = backend.matrix_inverse_pth_root_batched(
              [statistics[indices[k]] for k in dense], [exponents[indices[k]] for k in dense],
              [sizes[indices[k]] for k in dense], ridge_epsilon=matrix_epsilon,
              relative_matrix_epsilon=relative_matrix_epsilon, eigh=eigh,
              out=[outs[k] for k in dense])
          rows[dense, :comm.METRICS_STRIDE] = m
        fd_calls, fd_slots = [], []
        lr_calls, lr_slots = [], []
        for k, i in enumerate(indices):
          if k in dense:
            continue
          if frequent_directions:
            # The reference packs the sketch at the PADDED size (every statistic is
            # padded to max_size, DS:2841-2843) and then crops to the block's own
            # rows (DS:2950): for blocks smaller than max_size the deflated
            # eigenvalues / has_zeros entries, which live in the last rows
            # (DS:587-591), fall outside the crop.  Reproduced here so that the
            # optimizer trajectory is identical.
            n_i, stat_i, prev_i = sizes[i], statistics[i], prev_for(i)
            if n_i < max_size:
              gp = torch.zeros((max_size, max_size), dtype=torch.float32, device=stat_i.device)
              gp[:n_i, :n_i] = stat_i
              pp = torch.zeros((max_size, prev_i.shape[1]), dtype=torch.float32,
                               device=stat_i.device)
              pp[:n_i] = prev_i
              stat_i, prev_i = gp, pp
            fd_calls.append(dict(
                new_grad=stat_i, p=exponents[i], rank=compression_rank,
                ridge_epsilon=matrix_epsilon,
                relative_matrix_epsilon=relative_matrix_epsilon, decay=beta2,
                padding_start=n_i, prev=prev_i, new_grad_is_gram=True,
                generate_training_metrics=generate_training_metrics,
                generate_fd_metrics=generate_fd_metrics))
            fd_slots.append((k, n_i))
          else:
            lr_calls.append(dict(
                matrix=statistics[i], p=exponents[i], compression_rank=compression_rank,
                ridge_epsilon=matrix_epsilon,
                relative_matrix_epsilon=relative_matrix_epsilon, padding_start=sizes[i]))
            lr_slots.append(k)
        if lr_calls:  # all low-rank roots of this rank in one batched call
          if hasattr(backend, "low_rank_root_batched"):
            results = backend.low_rank_root_batched(lr_calls)
          else:
            results = [backend.low_rank_root(c.pop("matrix"), c.pop("p"), **c) for c in lr_calls]
          for k, (val, tm) in zip(lr_slots, results):
            outs[k].copy_(val)
            rows[k, 0] = tm.inverse_pth_root_errors
        if fd_calls:  # the eigen-step of all sketch updates of this rank runs batched
          for (k, n_i), (val, tm) in zip(fd_slots, backend.fd_update_root_batched(fd_calls)):
            outs[k].copy_(val[:n_i])
            rows[k, 0] = tm.inverse_pth_root_errors
            if fd_cols:  # FDDiagnostics ride behind the 8 PS_M_* columns of the gathered row
              rows[k, comm.METRICS_STRIDE:] = torch.stack(
                  [torch.as_tensor(getattr(tm.fd, name), dtype=torch.float32).to(rows.device)
                   for name in _FD_FIELDS])
        return rows

    root_fn = backend.matrix_inverse_pth_root_batched
    # Newton iteration counts of the PREVIOUS recompute, from the state (DS:338-351; zeros before
    # the first one = no hint): they steer the per-block accuracy policy of the root call
    # (ps_options.iters_hint) and weight the "lpt" ownership.  One small D2H per recompute.
    iters_hint = None
    if (iteration_count_hint and generate_training_metrics and not eigh and compression_rank == 0 and
        not lobpcg_topk_precondition):
      its = [st_.training_metrics.inverse_pth_root_iters for st_, num in zip(states, counts)
             if num > 0]
      if its and all(isinstance(t_, torch.Tensor) and t_.numel() > 0 for t_ in its):
        flat = torch.cat([t_.reshape(-1).to(torch.float32) for t_ in its])
        if flat.numel() == len(sizes):
          iters_hint = flat.cpu().tolist()
    # eigh path, eigh_solver="accurate" (the only solver with a keep rule in root calls): the condition number
    # every block had at the LAST recompute (metrics column 7 of the eigh rows).  Blocks far above the keep
    # rule's bound skip the fast path's attempt this time -- same bits (a hand-over starts the Jacobi solvers
    # from scratch), minus the attempt's time.  The memo lives in the optimizer object, not in the state: a
    # stale entry changes which of the two solvers roots a block (both within the Jacobi solvers' accuracy) or
    # costs the attempt's time; a run resumed from a checkpoint starts with an empty memo, so ITS "accurate"
    # roots can differ from the uninterrupted run's in the last digits (iteration_count_hint=False for
    # bit-reproducible restarts).  The default solver ("auto") keeps every block's fast-path result and
    # neither reads nor needs the memo: its roots depend on the statistics alone.
    eigh_skip = None
    if (eigh and iteration_count_hint and eigh_solver == "accurate" and compression_rank == 0 and
        not lobpcg_topk_precondition and _backend_for_testing is None):
      memo = _eigh_memo
      if memo["cond"] is not None and len(memo["cond"]) == len(sizes):
        eigh_skip = memo["cond"]
      elif len(klens) == len(sizes):
        # no recompute yet: a statistic that has seen fewer gradient columns than it has rows is rank
        # deficient up to its epsilon (the usual state of the first recomputes): no attempt.  A
        # heuristic that costs time only: a block that has seen zero gradients (frozen parameter) is
        # epsilon I, perfectly conditioned, and still goes to the Jacobi solvers once.
        updates = int(step) // max(int(statistics_compute_steps), 1) + 1
        eigh_skip = [float("inf") if updates * k < n else 0.0 for k, n in zip(klens, sizes)]
    if lobpcg_topk_precondition:
      # top-k deflated roots (DS:787-812, 889-928); blocks not larger than k (possible
      # here because nothing is padded to max_size) take the plain iteration
      def root_fn(mats, exps_, pads_, ridge_epsilon=1e-6, relative_matrix_epsilon=True,
                  eigh=False, out=None, **_):
        big = [j for j, n_ in enumerate(pads_) if n_ > lobpcg_topk_precondition + 1]
        small = [j for j in range(len(mats)) if j not in big]
        rows = torch.zeros((len(mats), comm.METRICS_STRIDE), dtype=torch.float32,
                           device=mats[0].device)
        if big:
          _, m_big, _ = backend.matrix_inverse_pth_root_deflated_batched(
              [mats[j] for j in big], [exps_[j] for j in big], [pads_[j] for j in big],
              topk=lobpcg_topk_precondition, max_iter=lobpcg_max_iter,
              ridge_epsilon=ridge_epsilon, relative_matrix_epsilon=relative_matrix_epsilon,
              out=[out[j] for j in big])
          rows[big] = m_big
        if small:
          _, m_small = backend.matrix_inverse_pth_root_batched(
              [mats[j] for j in small], [exps_[j] for j in small], [pads_[j] for j in small],
              ridge_epsilon=ridge_epsilon, relative_matrix_epsilon=relative_matrix_epsilon,
              out=[out[j] for j in small])
          rows[small] = m_small
        return out, rows

    payload_elems = None
    if quantize_second_moment:
      # DS:3102-3127: the owner quantizes its roots and the all-gather carries int16 codes
      # + float32 diagonal + float32 bucket sizes (half the bytes of float32 roots).
      # Payload of statistic i, in float32 words, every part 16-byte aligned:
      #   [codes: n*n int16 | diagonal: n | bucket_size: n]
      def _parts(n):
        return (n * n + 1) // 2 + (-((n * n + 1) // 2) % 4), n + (-n % 4)

      payload_elems = [_parts(n)[0] + 2 * _parts(n)[1] for n in sizes]

      def _unpack(flat, n):
        c, v = _parts(n)
        codes = flat[:c].view(torch.int16)[:n * n].view(n, n)
        return codes, flat[c:c + n], flat[c + v:c + v + n]

      def compute_fn(indices, outs):  # _quantized_matrix_inverse_pth_root_vmap, DS:2746-2773
        tmp = [torch.empty((sizes[i], sizes[i]), dtype=torch.float32,
                           device=statistics[0].device) for i in indices]
        _, m = root_fn(
            [statistics[i] for i in indices], [exponents[i] for i in indices],
            [sizes[i] for i in indices], ridge_epsilon=matrix_epsilon,
            relative_matrix_epsilon=relative_matrix_epsilon, eigh=eigh, out=tmp,
            options=dict(root_options, **({"iters_hint": [eigh_skip[i] for i in indices]} if (eigh and eigh_skip is not None)
                                          else {} if iters_hint is None or eigh else {
                "iters_hint": [iters_hint[i] for i in indices]})))
        backend.quantize_grouped(tmp, qdt_second_moment, True,
                                 out=[_unpack(o, sizes[i]) for o, i in zip(outs, indices)])
        return m

    roots, metrics = comm.sharded_inverse_pth_roots(
        statistics, exponents, group=group, ridge_epsilon=matrix_epsilon,
        relative_matrix_epsilon=relative_matrix_epsilon, eigh=eigh,
        ownership=block_ownership, root_fn=root_fn, out_cols=out_cols,
        compute_fn=compute_fn, payload_elems=payload_elems, sizes=sizes,
        pi_first=(_backend_for_testing is None and not lobpcg_topk_precondition),
        metrics_cols=metrics_cols if compute_fn is not None else comm.METRICS_STRIDE,
        iters_hint=iters_hint, hint_in_ownership=not shard_stats, eigh_skip_hint=eigh_skip,
        options=(root_options if (compute_fn is None and not lobpcg_topk_precondition) else None))
    mhost = metrics[:, :comm.METRICS_STRIDE].detach().cpu().numpy()   # one small D2H per recompute
    errors = mhost[:, 0]
    if eigh and eigh_solver == "accurate" and _backend_for_testing is None:
      _eigh_memo["cond"] = [float(v) for v in mhost[:, 7]]
    if quantize_second_moment:
      roots = [QuantizedValue(*_unpack(r, n), qdt_second_moment, True, [n, n])
               for r, n in zip(roots, sizes)]
    new_p = []
    for i, (root, old) in enumerate(zip(roots, prev)):
      err = errors[i]
      bad = np.isnan(err) or err >= inverse_failure_threshold  # DS:2936-2943
      if bad and isinstance(old, torch.Tensor) and old._base is not None:
        # `old` is a view into the PREVIOUS recompute's gathered buffer: keeping the view
        # would pin that whole buffer (every preconditioner of the model) for as long as
        # this block keeps failing, and torch.save would serialise it.  Detach it.
        old = old.clone()
      new_p.append(old if bad else root)

    out, idx = [], 0
    for state, num in zip(states, counts):
      if num == 0:
        out.append(ParameterStats(
            state.diagonal_statistics, state.statistics, [], state.diagonal_momentum,
            state.momentum, state.avg_grad,
            init_training_metrics(0, generate_training_metrics,
                                  device=metrics.device)))
        continue
      tm = MaskedNode()
      if generate_training_metrics:
        m = metrics[idx:idx + num]
        tm = state.training_metrics.replace(
            inverse_pth_root_errors=m[:, 0].clone(),
            inverse_pth_root_iters=m[:, 1].clone(),
            final_error_ratio=m[:, 2].clone(),
            max_eigen_value=m[:, 3].clone(),
            total_retries=m[:, 4].clone())
        if fd_cols and m.shape[1] > comm.METRICS_STRIDE:
          tm = tm.replace(fd=FDDiagnostics(**{
              name: m[:, comm.METRICS_STRIDE + j].clone() for j, name in enumerate(_FD_FIELDS)}))
      out.append(ParameterStats(state.diagonal_statistics, state.statistics,
                                new_p[idx:idx + num], state.diagonal_momentum,
                                state.momentum, state.avg_grad, tm))
      idx += num
    return out

  # ---------------------------------------------------------------------------
  def _preconditioned_grads_all(grads_flat, states, params_flat):
    """Preconditioner.preconditioned_grad (DS:1645-1708) for the whole tree.  Blocks
    of parameters whose merged shape is 1-D or 2-D (the common case) are applied in
    two grouped launches — X_b = g_b^T P_L for every block, then Y_b = X_b^T P_R
    written straight into the merged gradient — instead of two products, a
    transpose and a concatenation per block.  Everything else takes the per-block
    path of blocking.Preconditioner."""
    out = [None] * len(grads_flat)
    plan = _tree_plan(params_flat)
    if plan is not None and all(g.is_contiguous() for g in grads_flat):
      precs_flat = [p for st in states for p in st.preconditioners]
      if quantize_second_moment:  # _maybe_dequantize_preconditioners, DS:2097-2106
        precs_flat = _to_float_many(precs_flat)
      if len(precs_flat) == len(plan.stat_dims) and all(
          p.is_contiguous() and p.dtype == torch.float32 for p in precs_flat):
        for i, (g, sk) in enumerate(zip(grads_flat, plan.skipped)):
          if not sk:
            out[i] = torch.empty_like(g)
        # (dequantized int16 preconditioners are column-scaled, not exactly symmetric: they keep
        # the reference's operand orientation)
        # Precondition of the symmetric operand layout (P_L g applied as P_L^T g): float32 roots of
        # this library's Newton / eigh kernels on statistics of its own Gram kernel are bitwise
        # symmetric (mirrored tiles; W W^T).  The LOBPCG-deflated roots add a rank-k correction
        # outside those kernels and keep the reference's orientation.  A state imported from another
        # implementation may hold statistics / preconditioners that are symmetric only to rounding:
        # P_L^T g then differs from the reference's g^T P orientation by that rounding (PS_APPLY_SYM=0
        # under PS_DEV_ENV=1 restores the reference's orientation for such a run).
        plan.apply_preconditioners(grads_flat, precs_flat, out,
                                   symmetric_precs=(not quantize_second_moment and
                                                    not lobpcg_topk_precondition))
        return out
    stage_a, stage_b, keep = [], [], []
    precs = [st.preconditioners for st in states]
    if quantize_second_moment:  # _maybe_dequantize_preconditioners, DS:2097-2106
      flat = _to_float_many([q for st in states for q in st.preconditioners])
      precs, k = [], 0
      for st in states:
        precs.append(flat[k:k + len(st.preconditioners)])
        k += len(st.preconditioners)
    for idx, (grad, state, param) in enumerate(zip(grads_flat, states, params_flat)):
      if _skip_preconditioning(param):
        continue
      pc = preconditioner_from_params(param)
      tshape = pc._transformed_shape
      should = pc.should_precondition_dims()
      if compression_rank != 0 or not all(should) or len(tshape) not in (1, 2):
        out[idx] = pc.preconditioned_grad(grad, precs[idx],
                                          tensordot_fn=backend.tensordot_axis0,
                                          matmul_fn=backend.matmul)
        continue
      g_t = grad.reshape(tshape)
      res = torch.empty(tshape, dtype=torch.float32, device=grad.device)
      nd = len(tshape)
      for i, (gb, ob) in enumerate(zip(pc._partitioner.partition(g_t),
                                       pc._partitioner.partition(res))):
        pcs = precs[idx][i * nd:(i + 1) * nd]
        if nd == 1:
          d = gb.shape[0]
          gb_c = gb.contiguous()
          keep.append(gb_c)
          stage_a.append((gb_c.view(d, 1), pcs[0], ob.view(1, d), True, False))
        else:
          m, n = gb.shape
          x = torch.empty((n, m), dtype=torch.float32, device=grad.device)
          stage_a.append((gb, pcs[0], x, True, False))   # [n, m] = g^T P_L
          stage_b.append((x, pcs[1], ob, True, False))   # [m, n] = X^T P_R
      out[idx] = res.reshape(tuple(param.shape))
    backend.gemm_grouped(stage_a)
    backend.gemm_grouped(stage_b)
    del keep
    return out

  def _transform_grad(grad, state, param, step, precond_grad=None):
    """DS:3496-3625: grafting, preconditioning, momentum."""
    pc = preconditioner_from_params(param)
    sgd_update = grad
    new_diag = state.diagonal_statistics.to_float()
    norm = torch.linalg.vector_norm
    if graft_type in (GraftingType.ADAGRAD, GraftingType.ADAGRAD_NORMALIZED):
      scaled = grad
      if graft_type == GraftingType.ADAGRAD_NORMALIZED:
        scaled = grad / (norm(grad) + _EPSILON)
      new_diag = state.diagonal_statistics.to_float() + torch.square(scaled)
      grafting_update = scaled / (torch.sqrt(new_diag) + diagonal_epsilon)
    elif graft_type in (GraftingType.RMSPROP, GraftingType.RMSPROP_NORMALIZED):
      scaled = grad
      if graft_type == GraftingType.RMSPROP_NORMALIZED:
        scaled = grad / (norm(grad) + _EPSILON)
      w1 = beta2
      w2 = beta2 if beta2 == 1.0 else 1.0 - beta2
      new_diag = w1 * state.diagonal_statistics.to_float() + w2 * torch.square(scaled)
      rms = scaled / (torch.sqrt(new_diag) + diagonal_epsilon)
      if clip_by_scaled_gradient_norm:
        scaled_norm = norm(rms) / float(np.sqrt(float(rms.numel())))
        rms = rms / torch.clamp(scaled_norm / clip_by_scaled_gradient_norm, min=1.0)
      grafting_update = rms
    elif graft_type in (GraftingType.SGD, GraftingType.NONE):
      grafting_update = sgd_update
    else:  # SQRT_N
      grafting_update = torch.ones_like(sgd_update) * torch.sign(sgd_update)

    lr = learning_rate(step) if callable(learning_rate) else learning_rate
    grafting_update = grafting_update * (lr if not decoupled_learning_rate else 1.0)

    if not _skip_preconditioning(param):
      if precond_grad is None:
        precond_grad = pc.preconditioned_grad(grad, state.preconditioners,
                                              tensordot_fn=backend.tensordot_axis0,
                                              matmul_fn=backend.matmul)
    else:
      if graft_type == GraftingType.NONE:
        logging.error("skipping preconditioning without grafting for param %s", param)
      precond_grad = grafting_update

    if graft_type is not GraftingType.NONE:
      multiplier = norm(grafting_update) / (norm(precond_grad) + _EPSILON)
    else:
      multiplier = 1.0
    shampoo_update = precond_grad * multiplier

    shampoo_wd, grafting_wd = shampoo_update, grafting_update
    if weight_decay != 0 and not decoupled_weight_decay:
      shampoo_wd = shampoo_update + weight_decay * param
      grafting_wd = grafting_update + weight_decay * param

    w = (1.0 - beta1) if moving_average_for_momentum else 1.0
    shampoo_mom = state.momentum.to_float() * beta1 + w * shampoo_wd
    grafting_mom = state.diagonal_momentum.to_float() * beta1 + w * grafting_wd

    run_shampoo = 1.0 if step >= start_preconditioning_step else 0.0
    momentum_update = run_shampoo * shampoo_mom + (1.0 - run_shampoo) * grafting_mom
    wd_update = run_shampoo * shampoo_wd + (1.0 - run_shampoo) * grafting_wd

    nesterov_update = momentum_update
    if nesterov:
      nesterov_update = w * wd_update + beta1 * momentum_update
    if weight_decay != 0 and decoupled_weight_decay:
      wd_lr = 1.0 if decoupled_learning_rate else lr
      nesterov_update = nesterov_update + wd_lr * weight_decay * param

    transformed = -1.0 * (lr if decoupled_learning_rate else 1.0) * nesterov_update
    new_state = ParameterStats(_quantize(new_diag), state.statistics,
                               state.preconditioners, _quantize(grafting_mom),
                               _quantize(shampoo_mom), state.avg_grad,
                               state.training_metrics)
    return transformed, new_state

  def _recompute_step(step):
    """Does _compute_preconditioners root on this step (DS:3460-3474)?"""
    steps_t = preconditioning_compute_steps
    if (decay_preconditioning_compute_steps and end_preconditioning_compute_steps
        and callable(learning_rate)):
      steps_t = preconditioning_compute_steps_schedule(
          learning_rate, preconditioning_compute_steps,
          end_preconditioning_compute_steps, step)
    return step % steps_t == 0

  def _transform_cfg(step):
    lr = learning_rate(step) if callable(learning_rate) else learning_rate
    return dict(
        graft_type=int(graft_type), nesterov=int(bool(nesterov)),
        moving_average_for_momentum=int(bool(moving_average_for_momentum)),
        decoupled_learning_rate=int(bool(decoupled_learning_rate)),
        decoupled_weight_decay=int(bool(decoupled_weight_decay)),
        run_shampoo=int(step >= start_preconditioning_step),
        beta1=float(beta1), beta2_w1=float(beta2),
        beta2_w2=float(beta2 if beta2 == 1.0 else 1.0 - beta2),
        diagonal_epsilon=float(diagonal_epsilon), weight_decay=float(weight_decay),
        lr=float(lr),
        clip_by_scaled_gradient_norm=float(clip_by_scaled_gradient_norm or 0.0))

  _donated = {}   # tree shapes -> plan.DonatedStep

  def _donated_update(grads_flat, stats_flat, params_flat, step):
    """update() on a donated state (steps without a root recompute): the updates, or None when
    this step / state takes the functional path."""
    if (_backend_for_testing is not None or quantize_second_moment or
        best_effort_memory_usage_reduction or lobpcg_topk_precondition or _recompute_step(step)):
      return None
    plan = _tree_plan(params_flat)
    if plan is None or not stats_flat:
      return None
    ds = _donated.get(id(plan))
    if ds is None:
      from .plan import DonatedStep
      ds = _donated[id(plan)] = DonatedStep(plan, _graft_type_has_diagonal_statistics(),
                                            weight_decay != 0)
    if not ds.bound_to(stats_flat):
      ok = (all(isinstance(st, ParameterStats) and
                st.momentum.quantized_dtype == torch.float32 and
                st.diagonal_momentum.quantized_dtype == torch.float32 and
                isinstance(st.avg_grad, MaskedNode) for st in stats_flat) and
            all(g.dtype == torch.float32 and g.is_contiguous() for g in grads_flat))
      if not ok or not ds.bind(stats_flat, grads_flat, params_flat, symmetric_precs=True):
        return None
    if not ds.qualifies(grads_flat, params_flat):   # every step: strides / dtype / device / shape
      return None
    do_stats = statistics_compute_steps <= 1 or step % statistics_compute_steps == 0
    w2 = beta2 if beta2 == 1.0 else 1.0 - beta2
    return ds.step(grads_flat, params_flat, _transform_cfg(step), do_stats, beta2, w2)

  def _transform_grads_fused(grads_flat, states, params_flat, pgs, step):
    """_transform_grad (DS:3496-3625) for the whole tree in three HIP launches."""
    lr = learning_rate(step) if callable(learning_rate) else learning_rate
    has_diag = _graft_type_has_diagonal_statistics()
    items = []
    for g, s, p, pg in zip(grads_flat, states, params_flat, pgs):
      skipped = _skip_preconditioning(p)
      if skipped and graft_type == GraftingType.NONE:
        logging.error("skipping preconditioning without grafting for param %s", p)
      items.append(dict(
          grad=g, pgrad=None if skipped else pg,
          param=p if weight_decay != 0 else None,
          diag_in=s.diagonal_statistics.to_float() if has_diag else None,
          mom_in=s.momentum.to_float(), dmom_in=s.diagonal_momentum.to_float()))
    cfg = dict(
        graft_type=int(graft_type), nesterov=int(bool(nesterov)),
        moving_average_for_momentum=int(bool(moving_average_for_momentum)),
        decoupled_learning_rate=int(bool(decoupled_learning_rate)),
        decoupled_weight_decay=int(bool(decoupled_weight_decay)),
        run_shampoo=int(step >= start_preconditioning_step),
        beta1=float(beta1), beta2_w1=float(beta2),
        beta2_w2=float(beta2 if beta2 == 1.0 else 1.0 - beta2),
        diagonal_epsilon=float(diagonal_epsilon), weight_decay=float(weight_decay),
        lr=float(lr),
        clip_by_scaled_gradient_norm=float(clip_by_scaled_gradient_norm or 0.0))
    res = backend.transform_grads_fused(items, cfg)
    outs = []
    for (upd, nd, mom, dmom), s in zip(res, states):
      new_diag = nd if has_diag else s.diagonal_statistics.to_float()
      outs.append((upd, ParameterStats(_quantize(new_diag), s.statistics, s.preconditioners,
                                       _quantize(dmom), _quantize(mom), s.avg_grad,
                                       s.training_metrics)))
    return outs

  # ---------------------------------------------------------------------------
  def update_fn(grads, state, params):
    """DS:3627-3659."""
    with _collector_paused():
      return _update_impl(grads, state, params)

  def _update_impl(grads, state, params):
    params_flat, treedef = pytree.tree_flatten(params)
    stats_flat = treedef.flatten_up_to(state.stats)
    grads_flat = treedef.flatten_up_to(grads)
    step = int(state.count)
    # Mixed-precision training hands over bf16 / fp16 gradients: the reference promotes
    # them inside its float32 arithmetic; the kernels here take float32 only, so promote
    # at the door and hand the updates back in the gradient's dtype.
    grad_dtypes = [g.dtype for g in grads_flat]
    if donate_state and all(dt == torch.float32 for dt in grad_dtypes):
      upd = _donated_update(grads_flat, stats_flat, params_flat, step)
      if upd is not None:   # state updated in place: the same objects go back
        return (treedef.unflatten(upd), ShampooState(count=state.count + 1, stats=state.stats))
    grads_flat = [g if g.dtype == torch.float32 else g.to(torch.float32) for g in grads_flat]
    # strided views (a `.t()` of a square buffer, an expanded tensor) are copied: the kernels address a
    # gradient block as base pointer + planned offsets of a contiguous tensor
    grads_flat = [g if g.is_contiguous() else g.contiguous() for g in grads_flat]
    if any(p.dtype != torch.float32 for p in params_flat):
      params_flat = [p if p.dtype == torch.float32 else p.to(torch.float32)
                     for p in params_flat]

    new_stats = _compute_stats_all(grads_flat, stats_flat, params_flat, step)
    new_stats = _compute_preconditioners(new_stats, params_flat, step)
    pgs = _preconditioned_grads_all(grads_flat, new_stats, params_flat)
    if best_effort_memory_usage_reduction:
      # int8 momentum (DS:3581-3586 to_float): dequantize the whole tree in one launch
      n = len(new_stats)
      fl = _to_float_many([s.momentum for s in new_stats] +
                          [s.diagonal_momentum for s in new_stats])
      new_stats = [s._replace(momentum=_quantize(fl[i]), diagonal_momentum=_quantize(fl[n + i]))
                   for i, s in enumerate(new_stats)]
    if hasattr(backend, "transform_grads_fused"):
      outs = _transform_grads_fused(grads_flat, new_stats, params_flat, pgs, step)
    else:  # host-logic test seam: the same arithmetic as torch elementwise ops
      outs = [_transform_grad(g, s, p, step, pg)
              for g, s, p, pg in zip(grads_flat, new_stats, params_flat, pgs)]
    updates_flat = [o[0] if o[0].dtype == dt else o[0].to(dt)
                    for o, dt in zip(outs, grad_dtypes)]
    new_stats = [o[1] for o in outs]
    if best_effort_memory_usage_reduction:  # _quantize_momentum, DS:3617-3619
      moms = _quantize_momentum_many([s.momentum.to_float() for s in new_stats], params_flat)
      dmoms = _quantize_momentum_many([s.diagonal_momentum.to_float() for s in new_stats],
                                      params_flat)
      new_stats = [s._replace(momentum=m, diagonal_momentum=d)
                   for s, m, d in zip(new_stats, moms, dmoms)]
    return (treedef.unflatten(updates_flat),
            ShampooState(count=state.count + 1, stats=treedef.unflatten(new_stats)))

  # ---------------------------------------------------------------------------
  # Sharded optimizer state: the reference's pjit mode (DS:2162-2583).  Same state layout
  # (ShardedShampooStats: GlobalShardedParameterStats with ONE stacked, padded array per kind
  # + a LocalShardedParameterStats per parameter) and the same init-function triple; the
  # XLA sharding of the stacked statistics along their leading axis becomes: rank r of the
  # process group keeps rows [r*b, (r+1)*b) (b = padded count / ranks — the batch() chunks
  # of DS:1827), updates and roots them in place at their TRUE sizes (views into the padded
  # slots, padding_start semantics), and one RCCL all-gather replicates the roots.
  def _pad_count(n_stats):
    world, _ = comm.world_and_rank(group)
    to_pad = -n_stats % world
    return (world if n_stats == 0 else n_stats + to_pad), world

  def _stat_sizes(params_flat):
    sizes_per_param, exps = [], []
    for param in params_flat:
      sizes = []
      if not _skip_preconditioning(param):
        pc = preconditioner_from_params(param)
        sizes = [int(sh[0]) for sh in pc.shapes_for_preconditioners()]
        e = pc.exponent_for_preconditioner() if exponent_override == 0 else exponent_override
        exps.extend([e] * len(sizes))
      sizes_per_param.append(sizes)
    return sizes_per_param, exps

  def _padded_eye(n, max_size, scale, dev):
    """pad_square_matrix(scale * I_n, max_size) (DS:1324-1350): [[scale I, 0], [0, I]]."""
    m = torch.eye(max_size, dtype=torch.float32, device=dev)
    if n > 0:
      m[:n, :n] *= scale
    return m

  def sharded_init_fn(params):
    """DS:2162-2255."""
    params_flat, treedef = pytree.tree_flatten(params)
    dev = params_flat[0].device if params_flat else None
    sizes_per_param, exps = _stat_sizes(params_flat)
    all_sizes = [n for sz in sizes_per_param for n in sz]
    max_size = max(all_sizes) if all_sizes else block_size
    n_pad, world = _pad_count(len(all_sizes))
    _, rank = comm.world_and_rank(group)
    b = n_pad // world
    pd = _precond_dim(compression_rank, max_size)
    local_flat, index = [], 0
    for param, sizes in zip(params_flat, sizes_per_param):
      zeros_q = _quantize_momentum_many([_zeros_f32_like(param), _zeros_f32_like(param)],
                                        [param, param])
      diag = _zeros_f32_like(param)   # DS:2211: always allocated in this mode, whatever the graft type
      local_flat.append(LocalShardedParameterStats(
          _quantize(diag), zeros_q[0], zeros_q[1],
          _zeros_f32_like(param) if (frequent_directions and average_grad) else MaskedNode(),
          init_training_metrics(len(sizes), generate_training_metrics, generate_fd_metrics,
                                device=dev),
          index, tuple(sizes)))
      index += len(sizes)
    # this rank's chunk of the stacked statistics; entries past the real ones are the
    # identity matrices the reference pads the list with (exponent 1, DS:2237-2249)
    mine = range(rank * b, (rank + 1) * b)
    stats = torch.stack([
        _padded_eye(all_sizes[i], max_size, matrix_epsilon, dev) if i < len(all_sizes)
        else torch.eye(max_size, dtype=torch.float32, device=dev) for i in mine])
    eye_p = torch.eye(max_size, pd, dtype=torch.float32, device=dev) * float(pd == max_size)
    preconds = eye_p.unsqueeze(0).repeat(n_pad, 1, 1)
    exponents = torch.tensor(exps + [1] * (n_pad - len(exps)), dtype=torch.int32, device=dev)
    return ShampooState(
        count=torch.zeros([], dtype=torch.int32),
        stats=ShardedShampooStats(GlobalShardedParameterStats(stats, preconds, exponents),
                                  treedef.unflatten(local_flat)))

  def sharded_init_partition_spec_fn(params, params_partition_spec,
                                     partition_spec_for_statistics):
    """DS:2275-2342: the state tree with the caller's partition specs in place of arrays
    (kept for API parity: checkpointing code walks it; nothing here consumes it)."""
    pspec_flat, _ = pytree.tree_flatten(params_partition_spec, is_leaf=lambda x: x is None)
    params_flat, treedef = pytree.tree_flatten(params)
    assert pspec_flat and params_flat
    sizes_per_param, _ = _stat_sizes(params_flat)
    local_flat, index = [], 0
    for param, pspec, sizes in zip(params_flat, pspec_flat, sizes_per_param):
      qdtype = quantized_dtype_for_momentum_buffers(param)
      scale = [] if qdtype == torch.float32 else (list(pspec[1:]) if pspec and len(pspec) > 1 else [])
      local_flat.append(LocalShardedParameterStats(
          QuantizedValue(pspec, [], [], torch.float32, False, list(param.shape)),
          QuantizedValue(pspec, [], scale, qdtype, False, list(param.shape)),
          QuantizedValue(pspec, [], scale, qdtype, False, list(param.shape)),
          pspec if (frequent_directions and average_grad) else MaskedNode(),
          pytree.tree_map(lambda _: None, init_training_metrics(
              1, generate_training_metrics, generate_fd_metrics)),
          index, tuple(sizes)))
      index += len(sizes)
    g = GlobalShardedParameterStats(partition_spec_for_statistics,
                                    partition_spec_for_statistics, None)
    return ShampooState(count=None, stats=ShardedShampooStats(g, treedef.unflatten(local_flat)))

  def sharded_init_shape_and_dtype_fn(params):
    """DS:2344-2418: the state tree with [shape, dtype] in place of arrays.  Shapes are the
    GLOBAL ones of the reference (statistics [N_padded, max, max]); a rank holds N_padded /
    ranks rows of the statistics."""
    params_flat, treedef = pytree.tree_flatten(params)
    sizes_per_param, _ = _stat_sizes(params_flat)
    all_sizes = [n for sz in sizes_per_param for n in sz]
    max_size = max(all_sizes) if all_sizes else block_size
    n_pad, _ = _pad_count(len(all_sizes))
    local_flat, index = [], 0
    for param, sizes in zip(params_flat, sizes_per_param):
      qdtype = quantized_dtype_for_momentum_buffers(param)
      shp = list(param.shape)
      scale = [] if qdtype == torch.float32 else [shp[1:], torch.float32]
      local_flat.append(LocalShardedParameterStats(
          QuantizedValue([shp, torch.float32], [], [], torch.float32, False, shp),
          QuantizedValue([shp, qdtype], [], scale, qdtype, False, shp),
          QuantizedValue([shp, qdtype], [], scale, qdtype, False, shp),
          [shp, param.dtype] if (frequent_directions and average_grad) else MaskedNode(),
          pytree.tree_map(lambda _: [[len(sizes)], torch.float32], init_training_metrics(
              len(sizes), generate_training_metrics, generate_fd_metrics)),
          index, tuple(sizes)))
      index += len(sizes)
    pd = _precond_dim(compression_rank, max_size)
    g = GlobalShardedParameterStats([[n_pad, max_size, max_size], torch.float32],
                                    [[n_pad, max_size, pd], torch.float32],
                                    [[n_pad], torch.int32])
    return ShampooState(count=[[], torch.float32],
                        stats=ShardedShampooStats(g, treedef.unflatten(local_flat)))

  def sharded_update_fn(grads, state, params):
    """DS:2420-2583."""
    with _collector_paused():
      return _sharded_update_impl(grads, state, params)

  def _sharded_update_impl(grads, state, params):
    params_flat, treedef = pytree.tree_flatten(params)
    grads_flat = treedef.flatten_up_to(grads)
    grad_dtypes = [g.dtype for g in grads_flat]
    grads_flat = [g if g.dtype == torch.float32 else g.to(torch.float32) for g in grads_flat]
    # strided views (a `.t()` of a square buffer, an expanded tensor) are copied: the kernels address a
    # gradient block as base pointer + planned offsets of a contiguous tensor
    grads_flat = [g if g.is_contiguous() else g.contiguous() for g in grads_flat]
    if any(p.dtype != torch.float32 for p in params_flat):
      params_flat = [p if p.dtype == torch.float32 else p.to(torch.float32) for p in params_flat]
    step = int(state.count)
    gstats = state.stats.global_stats
    local_flat = treedef.flatten_up_to(state.stats.local_stats)
    world, rank = comm.world_and_rank(group)
    b = int(gstats.statistics.shape[0])
    lo = rank * b
    # _convert_to_parameter_stats (DS:1762-1788): true-size views into the padded slots;
    # statistics another rank owns are empty placeholders
    stats_flat = []
    for loc in local_flat:
      st, pcs = [], []
      for j, n in enumerate(loc.sizes):
        i = loc.index_start + j
        st.append(gstats.statistics[i - lo, :n, :n] if lo <= i < lo + b
                  else torch.empty((0, 0), dtype=torch.float32, device=gstats.statistics.device))
        pcs.append(gstats.preconditioners[i, :n, :_precond_dim(compression_rank, n)])
      stats_flat.append(ParameterStats(loc.diagonal_statistics, st, pcs, loc.diagonal_momentum,
                                       loc.momentum, loc.avg_grad, loc.training_metrics))
    new_stats = _compute_stats_all(grads_flat, stats_flat, params_flat, step)
    # Order of DS:2443-2452: the gradient is transformed with the preconditioners the state
    # CAME IN with (the roots computed below reach the update of the NEXT step), unlike the
    # pmap path, which roots first (DS:3648-3650).  Pinned by tests/golden/e2e_sharded.npz.
    pgs = _preconditioned_grads_all(grads_flat, new_stats, params_flat)
    if best_effort_memory_usage_reduction:
      # int8 momentum (DS:3581-3586 to_float): dequantize the whole tree in one launch
      n_p = len(new_stats)
      fl = _to_float_many([s_.momentum for s_ in new_stats] +
                          [s_.diagonal_momentum for s_ in new_stats])
      new_stats = [s_._replace(momentum=_quantize(fl[i]), diagonal_momentum=_quantize(fl[n_p + i]))
                   for i, s_ in enumerate(new_stats)]
    if hasattr(backend, "transform_grads_fused"):
      outs = _transform_grads_fused(grads_flat, new_stats, params_flat, pgs, step)
    else:
      outs = [_transform_grad(g, s, p, step, pg)
              for g, s, p, pg in zip(grads_flat, new_stats, params_flat, pgs)]
    updates_flat = [o[0] if o[0].dtype == dt else o[0].to(dt) for o, dt in zip(outs, grad_dtypes)]
    stats_flat = [o[1] for o in outs]
    if best_effort_memory_usage_reduction:
      mq = _quantize_momentum_many(
          [s_.momentum.to_float() for s_ in stats_flat] +
          [s_.diagonal_momentum.to_float() for s_ in stats_flat], params_flat + params_flat)
      stats_flat = [s_._replace(momentum=mq[i], diagonal_momentum=mq[len(stats_flat) + i])
                    for i, s_ in enumerate(stats_flat)]
    new_stats = _compute_preconditioners(stats_flat, params_flat, step)
    # back into the stacked arrays (pad_square_matrix semantics: the padding of a slot is
    # the identity for statistics, DS:2461-2465, and zero for the roots, DST:367-398)
    new_statistics = gstats.statistics.clone()
    recomputed = any(ns.preconditioners is not os.preconditioners and
                     any(a is not c for a, c in zip(ns.preconditioners, os.preconditioners))
                     for ns, os in zip(new_stats, stats_flat))
    new_preconds = gstats.preconditioners
    if recomputed:
      new_preconds = torch.zeros_like(gstats.preconditioners)
    new_local = []
    for loc, ns in zip(local_flat, new_stats):
      for j, n in enumerate(loc.sizes):
        i = loc.index_start + j
        if lo <= i < lo + b and ns.statistics[j].numel():
          new_statistics[i - lo, :n, :n] = ns.statistics[j]
        if recomputed:
          p_ij = ns.preconditioners[j]
          new_preconds[i, :n, :p_ij.shape[1]] = p_ij
      new_local.append(LocalShardedParameterStats(
          ns.diagonal_statistics, ns.diagonal_momentum, ns.momentum, ns.avg_grad,
          ns.training_metrics, loc.index_start, loc.sizes))
    new_state = ShampooState(
        count=state.count + 1,
        stats=ShardedShampooStats(
            GlobalShardedParameterStats(new_statistics, new_preconds, gstats.exponents),
            treedef.unflatten(new_local)))
    return treedef.unflatten(updates_flat), new_state

  if shard_optimizer_states:
    # DS:3661-3673: init() returns the three init functions instead of a state
    def _init_fns(unused_params):
      return InitFnState(init_fn=sharded_init_fn, pspec_fn=sharded_init_partition_spec_fn,
                         shape_and_dtype_fn=sharded_init_shape_and_dtype_fn)

    return GradientTransformation(_init_fns, sharded_update_fn)
  return GradientTransformation(init_fn, update_fn)
