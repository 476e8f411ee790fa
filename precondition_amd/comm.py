"""Partition of independent statistics blocks over the GPUs of a node, and the
all-gather of the resulting preconditioners (RCCL over xGMI via
torch.distributed; backend "nccl" is RCCL on ROCm).

Replaces the reference's pmap sharding in _pmap_compute_preconditioners
(DS:2841-2879): there, every statistic is padded to max_size, the list is padded
to a multiple of the device count with identity / exponent 1 / padding_start 0
entries, ``batch()`` gives rank r the contiguous chunk [r*b, (r+1)*b), each
replica roots its chunk and ``jax.lax.all_gather`` returns everything in list
order.  Here the ownership rule is the same (so results land in the same list
order), but nothing is padded: each rank roots its own statistics at their true
sizes into one flat send buffer, and ONE all-gather of equal-sized flat buffers
(+ one of the [b, 8] metrics table) leaves every rank holding every root.  The
per-statistic results are views into the gathered buffer (no unpack copies).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import os

import numpy as np
import torch

METRICS_STRIDE = 8


def resolve_group(batch_axis_name):
  """None -> single device.  A ProcessGroup -> that group.  Any other truthy
  value (the reference passes the pmap axis name) -> the default group."""
  if not batch_axis_name:
    return None
  import torch.distributed as dist
  if isinstance(batch_axis_name, dist.ProcessGroup):
    return batch_axis_name
  if not dist.is_initialized():
    raise RuntimeError(
        "batch_axis_name is set but torch.distributed is not initialised; start "
        "one process per GPU (torchrun) and call init_process_group('nccl')")
  return dist.group.WORLD


def world_and_rank(group) -> Tuple[int, int]:
  if group is None:
    return 1, 0
  import torch.distributed as dist
  return dist.get_world_size(group), dist.get_rank(group)


def reference_ownership(num_statistics: int, world: int) -> List[int]:
  """owner[i] under the reference's batch(): pad the count to a multiple of
  `world`, rank r owns the contiguous chunk [r*b, (r+1)*b) (DS:2844, 2862, 1827)."""
  padded = num_statistics + (-num_statistics % world)
  b = padded // world
  return [i // b for i in range(num_statistics)]


def block_costs(sizes: Sequence[int], exponents: Sequence[int],
                iters_hint: Optional[Sequence[float]] = None) -> List[float]:
  """Cost model of one root: iterations x products per iteration x tiles of a symmetric product.
  c(p) products of T (T + 1) / 2 tiles of 128 x 128 x n each (T = ceil(n / 128): the product
  kernel's unit of work, so a 768^2 block costs 21 tile-products of K = 768 and a 1000^2 one 36
  of K = 1024), times the Newton iterations the block took at the PREVIOUS recompute when the
  caller has them (`iters_hint`, the optimizer state's inverse_pth_root_iters: ViT-B blocks take
  7-17) and a neutral 10 otherwise."""

  def c_of_p(p):
    return int(np.floor(np.log2(p))) + bin(int(p)).count("1") - 1 + 2 if p > 0 else 1

  out = []
  for i, (n, p) in enumerate(zip(sizes, exponents)):
    t = (int(n) + 127) // 128
    it = 10.0
    if iters_hint is not None:
      h = float(iters_hint[i])
      if h >= 1.0 and h == h:
        it = h
    out.append(it * c_of_p(p) * (t * (t + 1) / 2.0) * (t * 128.0))
  return out


def cost_balanced_ownership(sizes: Sequence[int], exponents: Sequence[int],
                            world: int,
                            iters_hint: Optional[Sequence[float]] = None) -> List[int]:
  """Longest-processing-time assignment on `block_costs` (perf mode; the gathered result order
  is unchanged because results are addressed by index).  `iters_hint` must be identical on every
  rank (it is when it comes from the gathered metrics table of the previous recompute)."""
  cost = block_costs(sizes, exponents, iters_hint)
  order = sorted(range(len(sizes)), key=lambda i: (-cost[i], i))
  load = [0.0] * world
  owner = [0] * len(sizes)
  for i in order:
    r = int(np.argmin(load))
    owner[i] = r
    load[r] += cost[i]
  return owner


def ownership_table(sizes: Sequence[int], exponents: Sequence[int], world: int,
                    ownership: str = "reference",
                    iters_hint: Optional[Sequence[float]] = None) -> List[int]:
  """owner[i] of every statistic; a pure function of shapes (and, for "lpt", of the previous
  recompute's gathered iteration counts), identical on every rank."""
  if ownership == "reference":
    return reference_ownership(len(sizes), world)
  if ownership == "lpt":
    return cost_balanced_ownership(sizes, exponents, world, iters_hint)
  raise ValueError(f"unknown ownership {ownership!r}")


_HI_STREAM = {}


def high_priority_stream(device):
  """ONE high-priority stream per device and process (the first phase of a two-phase exchange, bench.py's
  weak-scaling step): live streams share about four hardware queues, and every further one can put two of the
  eigh path's stream groups on one queue (measured: its cfg3 step 211 instead of 173 ms behind two extra streams)."""
  key = str(device)
  if key not in _HI_STREAM:
    _HI_STREAM[key] = torch.cuda.Stream(device=device, priority=-1)
  return _HI_STREAM[key]


_WORKER = []


def side_worker():
  """ONE persistent host thread per process for the call that runs beside the caller's (the second phase of the
  exchange).  Persistent, because the library keeps per-thread host state (pinned status rings, events) for the
  life of a thread: a new thread per step would allocate it anew every step."""
  if not _WORKER:
    from concurrent.futures import ThreadPoolExecutor
    _WORKER.append(ThreadPoolExecutor(max_workers=1, thread_name_prefix="ps-side"))
  return _WORKER[0]


def _collective_in_flight(delta: int) -> int:
  """Tells the library that an asynchronous RCCL collective was launched (+1) / has been waited
  for (-1): its kernels hold CUs on RCCL's stream, so the register-resident power iteration --
  whose workgroups spin on team mates and are launched at the capacity of an idle chip -- is not
  used while one is in flight (include/ps_api.h: ps_collective_in_flight).  Returns 1 if the
  count was changed."""
  from . import _lib
  _lib.lib().ps_collective_in_flight(int(delta))
  return 1


def sharded_inverse_pth_roots(
    statistics: Sequence[torch.Tensor],
    exponents: Sequence[int],
    group=None,
    ridge_epsilon: float = 1e-6,
    relative_matrix_epsilon: bool = True,
    eigh: bool = False,
    ownership: str = "reference",
    root_fn: Optional[Callable] = None,
    out_cols: Optional[Sequence[int]] = None,
    compute_fn: Optional[Callable] = None,
    overlap: bool = True,
    overlap_min_bytes: int = 32 << 20,
    payload_elems: Optional[Sequence[int]] = None,
    sizes: Optional[Sequence[int]] = None,
    pi_first: Optional[bool] = None,
    metrics_cols: int = METRICS_STRIDE,
    iters_hint: Optional[Sequence[float]] = None,
    options: Optional[dict] = None,
    hint_in_ownership: bool = True,
    eigh_skip_hint: Optional[Sequence[float]] = None,
) -> Tuple[List[torch.Tensor], torch.Tensor]:
  """Roots every statistic on its owner rank and all-gathers the results.

  Returns (roots in list order, metrics [num_statistics, 8]) on every rank.
  `root_fn(matrices, ps, padding_starts, out=..., **kw) -> (roots, metrics)`
  defaults to the HIP batched root; tests inject a CPU function to exercise the
  sharding over gloo.  `out_cols[i]` (default n_i) is the stored width of result i
  (rank-compressed preconditioners are [n, rank+2], DS:520-532); `compute_fn(
  indices, outs) -> metrics[len(indices), 8]` replaces the plain batched root when
  statistics need per-index treatment (low-rank / Frequent-Directions branch).
  `payload_elems[i]` (needs `compute_fn`): result i travels as an opaque payload of
  that many float32 words (the int16-quantized preconditioner + its diagonal and
  bucket sizes, DS:3102-3127); `outs` and the returned roots are then flat views.
  `sizes[i]` (default statistics[i].shape[0]): with owner-only statistics
  (`shard_statistics`) the entries this rank does not own are placeholders.
  `pi_first` (HIP Newton root only; default: on whenever two phases are used): in the
  two-phase layout the power iteration runs once over all of this rank's statistics, BEFORE
  the first all-gather is in flight, and each phase's root call gets its largest eigenvalues
  from it (`max_ev=`): halves of the 100 short launches would sit on the launch-latency floor,
  and the register-resident power iteration must not start under a collective that holds CUs
  on another stream (the library is told about asynchronous gathers through
  `ps_collective_in_flight` and uses its streaming kernels meanwhile).  Same kernels,
  bit-identical results.
  `metrics_cols` (needs `compute_fn`): width of the gathered metrics rows when the
  per-statistic diagnostics are wider than the 8 PS_M_* columns (FDDiagnostics).
  `iters_hint` (HOST floats, one per statistic, identical on every rank): the Newton iteration
  counts of the previous recompute (column 1 of the metrics this function returned then).  They
  weight the "lpt" ownership (unless `hint_in_ownership` is False: owner-only statistics need an
  ownership that never moves) and are handed to the root call as ps_options.iters_hint (well
  conditioned blocks skip the averaged M updates and the segmented accumulation).
  `options`: per-call modes of the HIP root (kernels._lib.make_options), e.g. the product
  arithmetic the factory's `precision` selects.
  """
  n_stats = len(statistics)
  world, rank = world_and_rank(group)
  sizes = [int(s) for s in sizes] if sizes is not None else [int(s.shape[0]) for s in statistics]
  if iters_hint is not None:
    iters_hint = [float(h) for h in iters_hint]
    if len(iters_hint) != n_stats:
      raise ValueError("iters_hint must hold one value per statistic")
  owner = ownership_table(sizes, exponents, world, ownership,
                          iters_hint if hint_in_ownership else None)
  hip_root = root_fn is None
  if root_fn is None:
    from . import kernels
    root_fn = kernels.matrix_inverse_pth_root_batched

  cols = [int(c) for c in out_cols] if out_cols is not None else list(sizes)
  elems = [sizes[i] * cols[i] for i in range(n_stats)]
  raw = payload_elems is not None
  if raw:
    if compute_fn is None:
      raise ValueError("payload_elems needs compute_fn")
    elems = [int(e) for e in payload_elems]
  dev = next((s.device for s in statistics if hasattr(s, 'device')), None)

  # Phases: with several ranks and enough work, every rank roots its statistics in
  # two halves so that the all-gather of the first half (RCCL's own stream,
  # async) runs under the Newton iterations of the second.  Which half a statistic
  # belongs to is derived from shapes alone, identically on every rank.
  total = [0] * world
  for i in range(n_stats):
    total[owner[i]] += elems[i]
  n_phases = 2 if (group is not None and overlap and max(total) * 4 >= overlap_min_bytes) else 1
  phase_of, seen = [0] * n_stats, [0] * world
  for i in range(n_stats):
    r = owner[i]
    phase_of[i] = 1 if (n_phases == 2 and seen[r] * 2 >= total[r]) else 0
    seen[r] += elems[i]

  # Offsets of every statistic inside its owner's flat buffer of its phase; all
  # ranks derive the same tables from shapes alone (no communication).  Every result starts on
  # a 4 KB boundary: the per-statistic preconditioners are VIEWS into the gathered buffer and are
  # read by every step's application products, whose 16-byte loads need aligned bases -- packed
  # back to back, one 197 x 197 statistic (155,236 bytes) left every later preconditioner of the
  # ViT-B tree 4 bytes off and the whole application on its guarded scalar-load path: 3.4 instead
  # of 2.5 ms per step (tools/dev_r4_apply_real.py).
  ALIGN = 1024   # float32 words
  offsets = [0] * n_stats
  slot = [0] * n_stats
  fill = [[0] * world for _ in range(n_phases)]
  count = [[0] * world for _ in range(n_phases)]
  for i in range(n_stats):
    r, ph = owner[i], phase_of[i]
    offsets[i] = fill[ph][r]
    fill[ph][r] += -(-elems[i] // ALIGN) * ALIGN
    slot[i] = count[ph][r]
    count[ph][r] += 1

  lam_of = None
  if pi_first is None:
    pi_first = n_phases == 2 and hip_root
  all_mine = [i for i in range(n_stats) if owner[i] == rank]
  pi_options = {k: v for k, v in (options or {}).items() if k in ("power_iteration", "pi_timeout_ms")}

  def _expired_waits():
    import ctypes
    from . import _lib
    n = ctypes.c_uint(0)
    _lib.lib().ps_power_iteration_health(ctypes.addressof(n), None, None)
    return n.value

  def _power_iteration():
    from . import kernels
    lam_, _ = kernels.power_iteration_batched([statistics[i] for i in all_mine],
                                              padding_starts=[sizes[i] for i in all_mine],
                                              options=pi_options or None)
    return lam_

  pi_expired_before = None
  if (pi_first and n_phases == 2 and compute_fn is None and not eigh and
      relative_matrix_epsilon and all_mine):
    # The standalone power iteration has no in-call recovery of an expired resident wait (the
    # root call's own power iteration has): the count is read here and again after the first
    # root call, which waits on the host for GPU progress and therefore sees the power iteration
    # finished; if it moved, `lam` holds NaN for the blocks of the team that gave up -- the
    # process is on the streaming kernels from then on, so the eigenvalues are recomputed and
    # the phase is rooted again.
    pi_expired_before = _expired_waits() if hip_root else None
    lam = _power_iteration()
    lam_of = {i: k for k, i in enumerate(all_mine)}
  gathered, gathered_metrics, handles = [], [], []
  in_flight = 0

  # ---- one phase = its send buffers, the root call of this rank's statistics in it, its all-gather --------------
  def _prepare(ph):
    buf_elems = max(max(fill[ph]), 1)
    max_count = max(max(count[ph]), 1)
    send = torch.empty((buf_elems,), dtype=torch.float32, device=dev)
    mine = [i for i in range(n_stats) if owner[i] == rank and phase_of[i] == ph]
    send_metrics = torch.zeros((max_count, metrics_cols), dtype=torch.float32, device=dev)
    outs = [send[offsets[i]:offsets[i] + elems[i]] for i in mine]
    if not raw:
      outs = [o.view(sizes[i], cols[i]) for o, i in zip(outs, mine)]
    return dict(buf_elems=buf_elems, max_count=max_count, send=send, mine=mine, send_metrics=send_metrics, outs=outs)

  def _root(P, lam_):
    mine, outs = P["mine"], P["outs"]
    extra = {}
    if options or iters_hint is not None or eigh_skip_hint is not None:
      o = dict(options or {})
      if iters_hint is not None and not eigh:
        o["iters_hint"] = np.asarray([iters_hint[i] for i in mine], dtype=np.float32)
      if eigh_skip_hint is not None and eigh:
        # the blocks' condition numbers at the last recompute: far above the keep rule = no attempt
        o["iters_hint"] = np.asarray([eigh_skip_hint[i] for i in mine], dtype=np.float32)
      extra["options"] = o
    kw = dict(extra)
    if lam_of is not None:
      kw["max_ev"] = lam_[[lam_of[i] for i in mine]]
    return root_fn([statistics[i] for i in mine], [exponents[i] for i in mine],
                   [sizes[i] for i in mine], ridge_epsilon=ridge_epsilon,
                   relative_matrix_epsilon=relative_matrix_epsilon, eigh=eigh,
                   out=outs, **kw)[1]

  def _gather(P, on_stream=None):
    """All-gather of a phase.  on_stream: the collectives are ordered behind (and issued under) that stream, but
    the receive buffers are allocated on the CALLER's stream, where the roots (views of them) are used afterwards:
    a block of another stream's pool could be handed out again while the caller's kernels still read it."""
    nonlocal in_flight
    buf_elems, max_count, send, send_metrics = P["buf_elems"], P["max_count"], P["send"], P["send_metrics"]
    if group is None:
      gathered.append(send.unsqueeze(0))
      gathered_metrics.append(send_metrics.unsqueeze(0))
      return
    import contextlib
    import torch.distributed as dist
    # flat outputs (concatenation form) are accepted by both RCCL and gloo
    g = torch.empty((world * buf_elems,), dtype=torch.float32, device=dev)
    gm = torch.empty((world * max_count * metrics_cols,), dtype=torch.float32, device=dev)
    with (torch.cuda.stream(on_stream) if on_stream is not None else contextlib.nullcontext()):
      _gather_into(P, g, gm, dist)
    gathered.append(g.view(world, buf_elems))
    gathered_metrics.append(gm.view(world, max_count, metrics_cols))

  def _gather_into(P, g, gm, dist):
    nonlocal in_flight
    send, send_metrics = P["send"], P["send_metrics"]
    if dist.get_backend(group) == "gloo" and send.is_cuda:
      # gloo has no device all-gather: stage through the host (functional fallback
      # for single-GPU debugging; the production backend is RCCL)
      g_h = torch.empty(g.shape, dtype=torch.float32)
      m_h = torch.empty(gm.shape, dtype=torch.float32)
      dist.all_gather_into_tensor(g_h, send.cpu(), group=group)
      dist.all_gather_into_tensor(m_h, send_metrics.reshape(-1).cpu(), group=group)
      g.copy_(g_h)
      gm.copy_(m_h)
    elif n_phases == 2 and send.is_cuda:
      in_flight += _collective_in_flight(+1)
      handles.append(dist.all_gather_into_tensor(g, send, group=group, async_op=True))
      handles.append(dist.all_gather_into_tensor(gm, send_metrics.reshape(-1), group=group,
                                                 async_op=True))
      handles.append((send, send_metrics))  # keep the send buffers alive until wait()
    else:
      dist.all_gather_into_tensor(g, send, group=group)
      dist.all_gather_into_tensor(gm, send_metrics.reshape(-1), group=group)

  # Two phases of the plain HIP Newton root run SIDE BY SIDE: the second phase's root call from a second host thread
  # on the caller's stream, the first on a high-priority stream, so that it finishes first and its all-gather runs
  # under the rest of the second; the two calls' kernels fill each other's tails and ramps (one after the other the
  # halves of a rank's share are each too small to fill the chip: bench.py's weak-scaling step 16.4 -> 14.6 ms).
  # Collectives are issued by THIS thread, in phase order, on every rank.  The library keeps its host state per
  # thread and releases the GIL while it waits for the GPU.
  side_by_side = False
  if (n_phases == 2 and hip_root and compute_fn is None and not eigh and lam_of is not None and
      dev is not None and dev.type == "cuda" and not os.environ.get("PS_SHARDED_SEQUENTIAL")):
    import torch.distributed as dist
    side_by_side = dist.get_backend(group) != "gloo"
  try:
    if side_by_side:
      P = [_prepare(0), _prepare(1)]
      cur = torch.cuda.current_stream(dev)
      hi = high_priority_stream(dev)
      ready = torch.cuda.Event()
      ready.record(cur)                      # behind the power iteration and whatever produced the statistics
      res, err = [None, None], []

      def _run(ph, stream):
        try:
          if P[ph]["mine"]:
            with torch.cuda.stream(stream):
              stream.wait_event(ready)
              res[ph] = _root(P[ph], lam)
        except Exception as e:  # pylint: disable=broad-except
          err.append(e)

      t2 = side_worker().submit(_run, 1, cur)
      _run(0, hi)
      redo = False
      if pi_expired_before is not None and not err:
        # (the first root call waited on the host for GPU progress: the power iteration has finished)
        redo = _expired_waits() != pi_expired_before
        pi_expired_before = None
      if redo or err:
        t2.result()
        if err:
          raise err[0]
        cur.wait_stream(hi)
        lam = _power_iteration()             # streaming execution now (ps_power_iteration_health)
        for ph in range(2):
          if P[ph]["mine"]:
            res[ph] = _root(P[ph], lam)
          if res[ph] is not None:
            P[ph]["send_metrics"][:len(P[ph]["mine"])] = res[ph]
          _gather(P[ph])
      else:
        if res[0] is not None:
          with torch.cuda.stream(hi):
            P[0]["send_metrics"][:len(P[0]["mine"])] = res[0]
        _gather(P[0], on_stream=hi)
        t2.result()
        if err:
          raise err[0]
        if res[1] is not None:
          P[1]["send_metrics"][:len(P[1]["mine"])] = res[1]
        _gather(P[1])
        cur.wait_stream(hi)
    else:
      for ph in range(n_phases):
        P = _prepare(ph)
        if P["mine"]:
          if compute_fn is not None:
            m = compute_fn(P["mine"], P["outs"])
          else:
            m = _root(P, lam if lam_of is not None else None)
            if pi_expired_before is not None:
              if (options or {}).get("execution") == "persistent" and statistics[0].is_cuda:
                # the persistent execution only enqueues: the resident power iteration (queued before
                # it) may still be running, and its expiry counter with it
                torch.cuda.current_stream(statistics[0].device).synchronize()
              if _expired_waits() != pi_expired_before:
                lam = _power_iteration()     # streaming execution now (ps_power_iteration_health)
                m = _root(P, lam)
              pi_expired_before = None       # checked once: later phases reuse the repaired `lam`
          P["send_metrics"][:len(P["mine"])] = m
        _gather(P)
  finally:
    # also on an exception from a root / compute call or a later all-gather: outstanding
    # gathers are waited for and the library's in-flight count is restored (a count left above
    # zero would keep the whole process off the resident power iteration)
    try:
      for h in handles:
        if hasattr(h, "wait"):
          h.wait()
    finally:
      while in_flight > 0:
        _collective_in_flight(-1)
        in_flight -= 1

  roots = [
      gathered[phase_of[i]][owner[i], offsets[i]:offsets[i] + elems[i]] for i in range(n_stats)
  ]
  if not raw:
    roots = [r.view(sizes[i], cols[i]) for i, r in enumerate(roots)]
  rows = [gathered_metrics[phase_of[i]][owner[i], slot[i]] for i in range(n_stats)]
  metrics = torch.stack(rows, dim=0)
  return roots, metrics
