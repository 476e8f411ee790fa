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


def cost_balanced_ownership(sizes: Sequence[int], exponents: Sequence[int],
                            world: int) -> List[int]:
  """Longest-processing-time assignment on cost c(p) * n^3 (perf mode; the
  gathered result order is unchanged because results are addressed by index)."""

  def c_of_p(p):
    return int(np.floor(np.log2(p))) + bin(int(p)).count("1") - 1 + 2 if p > 0 else 1

  order = sorted(range(len(sizes)),
                 key=lambda i: -(c_of_p(exponents[i]) * float(sizes[i]) ** 3))
  load = [0.0] * world
  owner = [0] * len(sizes)
  for i in order:
    r = int(np.argmin(load))
    owner[i] = r
    load[r] += c_of_p(exponents[i]) * float(sizes[i]) ** 3
  return owner


def ownership_table(sizes: Sequence[int], exponents: Sequence[int], world: int,
                    ownership: str = "reference") -> List[int]:
  """owner[i] of every statistic; a pure function of shapes, identical on every rank."""
  if ownership == "reference":
    return reference_ownership(len(sizes), world)
  if ownership == "lpt":
    return cost_balanced_ownership(sizes, exponents, world)
  raise ValueError(f"unknown ownership {ownership!r}")


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
  """
  n_stats = len(statistics)
  world, rank = world_and_rank(group)
  sizes = [int(s) for s in sizes] if sizes is not None else [int(s.shape[0]) for s in statistics]
  owner = ownership_table(sizes, exponents, world, ownership)
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
  # ranks derive the same tables from shapes alone (no communication).
  offsets = [0] * n_stats
  slot = [0] * n_stats
  fill = [[0] * world for _ in range(n_phases)]
  count = [[0] * world for _ in range(n_phases)]
  for i in range(n_stats):
    r, ph = owner[i], phase_of[i]
    offsets[i] = fill[ph][r]
    fill[ph][r] += elems[i]
    slot[i] = count[ph][r]
    count[ph][r] += 1

  lam_of = None
  if pi_first is None:
    pi_first = n_phases == 2 and hip_root
  if (pi_first and n_phases == 2 and compute_fn is None and not eigh and
      relative_matrix_epsilon):
    from . import kernels
    all_mine = [i for i in range(n_stats) if owner[i] == rank]
    if all_mine:
      lam, _ = kernels.power_iteration_batched([statistics[i] for i in all_mine],
                                               padding_starts=[sizes[i] for i in all_mine])
      lam_of = {i: k for k, i in enumerate(all_mine)}
  gathered, gathered_metrics, handles = [], [], []
  in_flight = 0
  for ph in range(n_phases):
    buf_elems = max(max(fill[ph]), 1)
    max_count = max(max(count[ph]), 1)
    send = torch.empty((buf_elems,), dtype=torch.float32, device=dev)
    mine = [i for i in range(n_stats) if owner[i] == rank and phase_of[i] == ph]
    send_metrics = torch.zeros((max_count, metrics_cols), dtype=torch.float32, device=dev)
    if mine:
      outs = [send[offsets[i]:offsets[i] + elems[i]] for i in mine]
      if not raw:
        outs = [o.view(sizes[i], cols[i]) for o, i in zip(outs, mine)]
      if compute_fn is not None:
        m = compute_fn(mine, outs)
      else:
        extra = {}
        if lam_of is not None:
          extra["max_ev"] = lam[[lam_of[i] for i in mine]]
        _, m = root_fn([statistics[i] for i in mine], [exponents[i] for i in mine],
                       [sizes[i] for i in mine], ridge_epsilon=ridge_epsilon,
                       relative_matrix_epsilon=relative_matrix_epsilon, eigh=eigh,
                       out=outs, **extra)
      send_metrics[:len(mine)] = m
    if group is None:
      gathered.append(send.unsqueeze(0))
      gathered_metrics.append(send_metrics.unsqueeze(0))
      continue
    import torch.distributed as dist
    # flat outputs (concatenation form) are accepted by both RCCL and gloo
    g = torch.empty((world * buf_elems,), dtype=torch.float32, device=dev)
    gm = torch.empty((world * max_count * metrics_cols,), dtype=torch.float32, device=dev)
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
    gathered.append(g.view(world, buf_elems))
    gathered_metrics.append(gm.view(world, max_count, metrics_cols))
  for h in handles:
    if hasattr(h, "wait"):
      h.wait()
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
