"""Low-rank ("Sketchy") preconditioner branch: rank-compressed roots and the
Frequent-Directions update (reference: _low_rank_root DS:1033-1120,
_fd_update_root DS:1123-1290, pack/unpack DS:540-592, frequent_directions_update
DS:1473-1505; BASELINE config 5).

Division of labour: everything O(d^2) or larger runs in the HIP library — the
Gram matrix (stats kernel), the covariance update C = decay * W W^T + R R^T
(gemm), its symmetric eigendecomposition (blocked Jacobi, ps_eigh_batched_f32; from
size 1024 only the leading rank+1 eigenpairs, by the block method of subspace.py on
the MFMA GEMM) and the error metrics (gemm).  What is left on the host are O(rank) vector
selections, masks and the exact packing layout, kept as torch index ops.

The reference takes an SVD of `updated = [sqrt(decay) * W | R]` (DS:1193); its
left singular vectors / squared singular values are the eigenpairs of
`updated @ updated.T`, which is what is decomposed here.  Singular subspaces are
only defined up to rotations inside clusters, so parity is checked on the
packed scalars (deflated eigs, tail, inverted eigs, const) and on the
reconstructed covariance, as the reference's own tests do (DST:770-885).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch

from . import kernels
from .blocking import _precond_dim, _should_compress
from .state import TrainingMetrics


# ---- packing (bit-exact layout, DS:555-592) -----------------------------------
def _fd_low_rank_unpack(preconditioner, compression_rank):
  r = abs(compression_rank)
  assert r != 0, compression_rank
  assert preconditioner.dim() == 2, preconditioner.shape
  dim, storage_dim = preconditioner.shape
  assert storage_dim < dim
  assert storage_dim == r + 2
  eigvecs = preconditioner[:, :r]
  inverted_eigvals = preconditioner[:r, -2]
  const = preconditioner[0, -1]
  eigvals = preconditioner[-r:, -1]
  tail = preconditioner[1, -1]
  has_zeros = preconditioner[-1, -2].bool()
  return eigvecs, eigvals, inverted_eigvals, const, tail, has_zeros


def _fd_low_rank_pack(eigvecs, deflated_eigs, inverted_eigs, new_const, new_tail,
                      has_zeros, rank):
  rank = abs(rank)
  assert rank > 0
  d = eigvecs.shape[0]
  assert eigvecs.shape[1] == rank and eigvecs.dim() == 2
  assert list(deflated_eigs.shape) == [rank], deflated_eigs.shape
  assert list(inverted_eigs.shape) == [rank], inverted_eigs.shape
  assert _precond_dim(rank, d) == rank + 2 and rank + 2 < d
  precond = torch.zeros((d, rank + 2), dtype=torch.float32, device=eigvecs.device)
  precond[:, :rank] = eigvecs
  precond[:rank, -2] = inverted_eigs
  precond[0, -1] = new_const
  precond[1, -1] = new_tail
  precond[-rank:, -1] = deflated_eigs
  precond[-1, -2] = torch.as_tensor(has_zeros).to(torch.float32)
  return precond


def _low_rank_unpack(preconditioner, compression_rank):
  eigvecs, _, inverted_eigvals, const, _, has_zeros = _fd_low_rank_unpack(
      preconditioner, compression_rank)
  return eigvecs, inverted_eigvals, const, has_zeros


def _low_rank_pack(eigvecs, eigvals, const, compression_rank):
  return _fd_low_rank_pack(eigvecs, torch.zeros_like(eigvals), eigvals, const, 0.0,
                           False, compression_rank)


def _metrics(error) -> TrainingMetrics:
  return TrainingMetrics(inverse_pth_root_errors=torch.as_tensor(
      error, dtype=torch.float32))


# ---- _low_rank_root (DS:1033-1120) ----------------------------------------------
def _low_rank_root(matrix: torch.Tensor, p: int, compression_rank: int = 0,
                   ridge_epsilon: float = 1e-6, error_tolerance: float = 1e-6,
                   relative_matrix_epsilon: bool = True,
                   padding_start: Optional[int] = None, prev=None
                   ) -> Tuple[torch.Tensor, TrainingMetrics]:
  """Top- (rank > 0) or bottom- (rank < 0) |rank| eigenpairs of the inverse p-th
  root plus the mean of the remaining inverted eigenvalues, packed [d, |rank|+2]."""
  del prev
  return _low_rank_root_batched([dict(
      matrix=matrix, p=p, compression_rank=compression_rank, ridge_epsilon=ridge_epsilon,
      error_tolerance=error_tolerance, relative_matrix_epsilon=relative_matrix_epsilon,
      padding_start=padding_start)])[0]


def _low_rank_root_batched(calls) -> List[Tuple[torch.Tensor, TrainingMetrics]]:
  """_low_rank_root for a list of keyword dicts: ONE batched power iteration, one batched
  eigendecomposition and two grouped products (the error metric) for all statistics of a
  recompute, and no host synchronisation — the largest eigenvalue and the ridge stay on
  the device (the reference runs this under vmap, DS:2742-2744)."""
  out: List[Optional[Tuple[torch.Tensor, TrainingMetrics]]] = [None] * len(calls)
  live = []
  for idx, c in enumerate(calls):
    matrix, rank = c["matrix"], c["compression_rank"]
    assert rank != 0
    d = matrix.shape[0]
    assert matrix.shape[0] == matrix.shape[1] and d > abs(rank) + 2
    ps = c.get("padding_start")
    real_dim = d if ps is None else int(ps)
    if real_dim == 0:  # DS:1114-1118
      out[idx] = (torch.zeros((d, abs(rank) + 2), dtype=torch.float32, device=matrix.device),
                  _metrics(0.0))
    else:
      live.append((idx, matrix[:real_dim, :real_dim].contiguous(), real_dim))
  if not live:
    return out
  dev = live[0][1].device
  mats = [a for _, a, _ in live]
  # largest eigenvalues for the relative ridge: all matrices in one call, result on the device
  rel = [j for j, (idx, _, _) in enumerate(live) if calls[idx].get("relative_matrix_epsilon", True)]
  max_ev = [None] * len(live)
  if rel:
    tols = {float(calls[live[j][0]].get("error_tolerance", 1e-6)) for j in rel}
    for tol in tols:   # the stop tolerance of the power iteration is per call
      grp = [j for j in rel if float(calls[live[j][0]].get("error_tolerance", 1e-6)) == tol]
      lam, _ = kernels.power_iteration_batched([mats[j] for j in grp], 100, tol)
      for k, j in enumerate(grp):
        max_ev[j] = lam[k]
  regs, ridges = [], []
  for j, (idx, a, real_dim) in enumerate(live):
    c = calls[idx]
    tol = float(c.get("error_tolerance", 1e-6))
    eps = float(c.get("ridge_epsilon", 1e-6))
    if max_ev[j] is None:
      ridge = torch.tensor(eps * max(1.0, tol), dtype=torch.float32, device=dev)
    else:
      ridge = eps * torch.clamp(max_ev[j], min=tol)
    ridges.append(ridge)
    regs.append(a + ridge * torch.eye(real_dim, dtype=torch.float32, device=dev))
  es, us = kernels.eigh_batched(regs)  # ascending, like LAPACK
  tmp = [torch.empty_like(r) for r in regs]
  rec = [torch.empty_like(r) for r in regs]
  kernels.gemm_grouped([(r, u, t, False, False) for r, u, t in zip(regs, us, tmp)])
  kernels.gemm_grouped([(u, t, x, True, False) for u, t, x in zip(us, tmp, rec)])
  for j, (idx, a, real_dim) in enumerate(live):
    c = calls[idx]
    rank, p = c["compression_rank"], c["p"]
    r = abs(rank)
    d = c["matrix"].shape[0]
    e, u, ridge = es[j], us[j], ridges[j]
    error = (rec[j] - torch.diag(e)).abs().max()
    alpha = -1.0 / p
    inv_e = torch.where(e == 0.0, torch.zeros_like(e), torch.clamp(e, min=ridge) ** alpha)
    # The reference's padded problem has (d - real_dim) zero eigenvalues in front
    # (inv_e = 0 there); after its flip/roll the kept pairs come first (DS:1085-1097).
    if rank < 0:
      order = torch.arange(real_dim, device=dev)            # [low .. hi]
    else:
      order = torch.arange(real_dim - 1, -1, -1, device=dev)  # [hi .. low]
    inv_sorted, u_sorted = inv_e[order], u[:, order]
    k = min(r, real_dim)
    keep_e = torch.zeros((r,), dtype=torch.float32, device=dev)
    u_keep = torch.zeros((d, r), dtype=torch.float32, device=dev)
    keep_e[:k] = inv_sorted[:k]
    u_keep[:real_dim, :k] = u_sorted[:, :k]
    num_avg = real_dim - r
    const = inv_sorted[k:].sum() / (num_avg if num_avg > 0 else 1.0)
    out[idx] = (_low_rank_pack(u_keep, keep_e, const, rank), _metrics(error))
  return out


# ---- frequent_directions_update (DS:1473-1505) -------------------------------------
def _gram_precision(d: int) -> str:
  """Arithmetic of the covariance update G G^T of a Frequent-Directions step, its own switch
  (PS_FD_GRAM = f32 | bf16x3): "bf16x3" (the default for aligned 2-D blocks of >= 1024 rows --
  BASELINE configs[4] names the bf16 MFMA for this branch) multiplies bf16 hi/lo pairs of g
  with float32 accumulation (2^-17 operand precision) and symmetrises the result; "f32" is the
  exact-f32 MFMA statistics kernel the dense optimizer always uses (the reference computes this
  factor at HIGHEST precision, DS:1469)."""
  import os
  mode = os.environ.get("PS_FD_GRAM", "bf16x3")
  if mode not in ("f32", "bf16x3"):
    raise ValueError(f"PS_FD_GRAM must be f32 or bf16x3, got {mode!r}")
  return mode


def gram_of_block(g: torch.Tensor, axis: int) -> torch.Tensor:
  """tensordot(g, g, all axes but `axis`): the covariance update of one Frequent-Directions
  step (see _gram_precision); everything that is not a large aligned 2-D block, and the dense
  statistics of the optimizer, always use the float32 MFMA statistics kernel."""
  d = g.shape[axis]
  if (g.dim() == 2 and d >= 1024 and d % 32 == 0 and g.shape[1 - axis] % 32 == 0 and
      _gram_precision(d) != "f32"):
    gt = g if axis == 0 else g.t()
    hi, lo = kernels.to_bf16(gt.contiguous() if not gt.is_contiguous() else gt, split=True)
    out = torch.empty((d, d), dtype=torch.float32, device=g.device)
    # the upper tile triangle only, every tile stored with its mirror image (hi*lo and lo*hi enter the
    # float32 accumulation chain in a fixed order, so a full product would be symmetric only up to
    # rounding: its consumers, the eigensolvers, want exact symmetry)
    kernels.gemm_bf16_grouped([((hi, lo), (hi, lo), out)], symmetric=True)
    return out
  zero = torch.zeros((d, d), dtype=torch.float32, device=g.device)
  out = torch.empty_like(zero)
  kernels.stats_update_grouped([(g, axis, zero, out)], 0.0, 1.0)
  return out


def frequent_directions_update(old_stats_factor, g: torch.Tensor, axis: int, w1, w2):
  """A square factor R with R R^T = tensordot(g, g) (old stats and weights are
  ignored, as in the reference).  The reference takes it from a QR; any factor
  with that property is equivalent downstream (only R R^T is ever used), so it is
  built here from the Gram matrix's eigendecomposition: R = U sqrt(max(e, 0))."""
  del old_stats_factor, w1, w2
  gram = gram_of_block(g, axis)
  (e,), (u,) = kernels.eigh_batched([gram])
  # significant columns first (descending), like the triangular factor: the
  # consumer masks factor COLUMNS >= padding_start (DS:1173-1174)
  e, u = torch.flip(e, dims=[0]), torch.flip(u, dims=[1])
  return u * torch.sqrt(torch.clamp(e, min=0.0))


# ---- _fd_update_root (DS:1123-1290) -------------------------------------------------
# Size from which the covariance update's leading eigenpairs come from the block method
# of subspace.py instead of the full eigendecomposition, and the width it may use.
SUBSPACE_MIN_N = 1024


def _fd_update_root(new_grad: torch.Tensor, p: int, rank: int = 0,
                    ridge_epsilon: float = 1e-6, error_tolerance: float = 1e-6,
                    relative_matrix_epsilon: bool = True, decay: float = 1.0,
                    padding_start: Optional[int] = None, prev: torch.Tensor = None,
                    generate_training_metrics: bool = False,
                    generate_fd_metrics: bool = False, new_grad_is_gram: bool = False
                    ) -> Tuple[torch.Tensor, TrainingMetrics]:
  """One Frequent-Directions sketch update.  `new_grad` is a factor R with
  R R^T = Gram (reference semantics) or, with new_grad_is_gram, the Gram itself."""
  return _fd_update_root_batched([dict(
      new_grad=new_grad, p=p, rank=rank, ridge_epsilon=ridge_epsilon,
      error_tolerance=error_tolerance, relative_matrix_epsilon=relative_matrix_epsilon,
      decay=decay, padding_start=padding_start, prev=prev,
      generate_training_metrics=generate_training_metrics,
      generate_fd_metrics=generate_fd_metrics, new_grad_is_gram=new_grad_is_gram)])[0]


def prepare_fd(d: int, rank: int, factors: int, device, p: int = 4, reps: int = 2) -> None:
  """Init-time preparation of the Frequent-Directions branch for `factors` factors of dimension
  `d` and sketch rank `rank` on `device`: `reps` throw-away sketch updates of random covariances
  of those shapes.  The first update of a process otherwise pays for loading the code objects of
  ~20 kernels and for growing the caching allocator's pool to the working set of an update (2 GB
  for 8 x 4096^2; each growth step is a hipMalloc that synchronises the device): 650 ms for 8 x
  4096^2 / rank 64, then 80 ms for the second one (a different allocation pattern once the sketch
  is non-zero), 26 ms from the third on.  The optimizer calls this from init_fn (outside any step);
  bench.py calls it before its timed FD updates.  Only sizes that take the block subspace path
  (d >= SUBSPACE_MIN_N on the GPU) need it."""
  if d < SUBSPACE_MIN_N or rank <= 0 or factors <= 0 or not torch.device(device).type == "cuda":
    return
  gen = torch.Generator(device=device).manual_seed(d + rank)
  prevs = [torch.zeros((d, rank + 2), dtype=torch.float32, device=device) for _ in range(factors)]
  for _ in range(max(1, reps)):
    calls = []
    for f in range(factors):
      g = torch.randn((d, d), generator=gen, device=device, dtype=torch.float32)
      calls.append(dict(new_grad=gram_of_block(g, 0), p=p, rank=rank, ridge_epsilon=1e-6, decay=0.999,
                        padding_start=d, prev=prevs[f], new_grad_is_gram=True))
      del g
    prevs = [r[0] for r in _fd_update_root_batched(calls)]
  torch.cuda.synchronize(device)


def _fd_group_key(kw):
  """Calls that can share the stacked fast path of _fd_update_root_group: no padding, no
  diagnostics, the block method applies, and identical scalars."""
  ng, prev, rank = kw["new_grad"], kw.get("prev"), kw.get("rank", 0)
  d = int(ng.shape[0])
  ps = d if kw.get("padding_start") is None else int(kw["padding_start"])
  want_fd = bool(kw.get("generate_fd_metrics") and kw.get("generate_training_metrics", True))
  if (prev is None or rank <= 0 or ps != d or want_fd or d < SUBSPACE_MIN_N or
      4 * (rank + 33) > d or os.environ.get("PS_FD_STACKED", "1") == "0"):
    return None
  return (d, rank, kw["p"], float(kw.get("decay", 1.0)), float(kw.get("ridge_epsilon", 1e-6)),
          float(kw.get("error_tolerance", 1e-6)), bool(kw.get("relative_matrix_epsilon", True)),
          bool(kw.get("new_grad_is_gram", False)), ng.device)


def _fd_update_root_group(calls, key):
  """_fd_update_root (DS:1123-1290) for factors that share their sizes and scalars: every
  O(d r) / O(r) step runs ONCE on tensors stacked over the factors (one launch per operation
  for the whole group, two host reads per call instead of three per factor), the arithmetic of
  _fd_prepare / _fd_finish term for term.  Returns None for a factor whose block iteration did
  not converge (the caller sends it through the general path)."""
  from . import subspace
  d, rank, p, decay, ridge_eps, err_tol, rel, is_gram, dev = key
  bsz, r = len(calls), rank
  prev = torch.stack([kw["prev"] for kw in calls])                       # [B, d, r + 2]
  # ONE library call for the whole update (ps_fd_update_batched_f32: preparation, every outer round of the
  # subspace iteration, DS:1196-1290 and the packing) where the fused kernels apply; the start block is the
  # seeded Gaussian block of subspace.top_eigenpairs_batched, so the eigenpairs are bit-identical to the
  # step-by-step path below (PS_FD_ONE_CALL=0 selects it; it is also the fallback for other shapes / modes).
  if (dev.type == "cuda" and os.environ.get("PS_FD_ONE_CALL", "1") != "0" and
      subspace._filter_precision(d) == "bf16x3" and subspace._fused_filter() and
      all(os.environ.get(k_, "1") != "0" for k_ in ("PS_FD_TILED", "PS_FD_ROUND_CALL", "PS_FD_FRAG", "PS_FD_RR_X6",
                                                     "PS_FD_PLANS", "PS_FD_ROUND_LIB", "PS_FD_CHOLQR"))):
    b = int(kernels.lib().ps_fd_block_columns(r, d))
    gen = torch.Generator(device=dev).manual_seed(1729)
    x0 = torch.randn((bsz, d, b), generator=gen, device=dev, dtype=torch.float32)
    res = kernels.fd_update_batched(
        [kw["new_grad"].contiguous() for kw in calls], prev.contiguous(), p, r, decay, ridge_eps, err_tol, rel, x0,
        input_is_factor=not is_gram, degree=int(os.environ.get("PS_FD_DEGREE", 12)))
    if res is not None:
      packed, conv, _ = res
      ok = conv.cpu().tolist()
      return [(packed[j], _metrics(0.0)) if ok[j] else None for j in range(bsz)]
  sketch = prev[:, :, :r]
  fwd_eigvals = prev[:, d - r:, -1]                                       # [B, r]
  tail = prev[:, 1, -1]                                                   # [B]
  if rel:
    ridge = ridge_eps * torch.clamp(fwd_eigvals[:, 0], min=err_tol)
  else:
    ridge = torch.full((bsz,), ridge_eps * max(1.0, err_tol), dtype=torch.float32, device=dev)
  weighted = (sketch * torch.sqrt(fwd_eigvals + ridge[:, None])[:, None, :]).contiguous()
  # C = decay * W W^T + R R^T (the SVD's u, s^2)
  c = torch.empty((bsz, d, d), dtype=torch.float32, device=dev)
  kernels.gemm_grouped([(weighted[j], weighted[j], c[j], False, True) for j in range(bsz)])
  grams = []
  for kw in calls:
    g = kw["new_grad"].contiguous()
    grams.append(g if is_gram else kernels.matmul(g, g, transb=True))
  kernels.fd_cov_update(c, grams, decay)      # c <- sym(decay * W W^T + gram), one pass
  e, u, conv, _ = subspace.top_eigenpairs_batched(list(c.unbind(0)), r + 1)   # e [B, r+1] desc
  # ---- _fd_finish, stacked ----
  noise = d * 1.2e-7 * torch.clamp(e.max(dim=1, keepdim=True).values, min=0.0)
  e = torch.where(e <= noise, torch.zeros_like(e), e)
  s_ = torch.sqrt(torch.clamp(e, min=0.0))
  cutoff = s_[:, r]
  rho_t = cutoff ** 2
  top_eigs = s_[:, :r]
  deflated = (top_eigs - cutoff[:, None]) * (top_eigs + cutoff[:, None])
  eigvecs = u[:, :, :r].clone()
  tail = tail * decay
  new_tail = tail + rho_t
  alpha = -1.0 / p
  new_const = torch.where(new_tail <= 0, torch.zeros_like(new_tail), new_tail ** alpha)
  new_tail = torch.where(new_tail <= 0, torch.zeros_like(new_tail), new_tail)
  deflated = torch.where(deflated <= 0, torch.zeros_like(deflated), deflated)
  eigvecs = eigvecs * (deflated > 0)[:, None, :]
  norms = torch.linalg.vector_norm(eigvecs, dim=1)
  safe = (0.99 <= norms) & (norms <= 1.01)
  eigvecs = eigvecs * safe[:, None, :]
  deflated = deflated * safe
  eigvecs = eigvecs / torch.where(safe, norms, torch.ones_like(norms))[:, None, :]
  upshifted = torch.square(top_eigs) + tail[:, None]
  upshifted = upshifted * (deflated > 0.0)
  upshifted = torch.where(upshifted <= 0, torch.zeros_like(upshifted), upshifted)
  inverted = torch.where(upshifted <= 0, torch.zeros_like(upshifted), upshifted ** alpha)
  has_zeros = (deflated <= 0).any(dim=1) | (new_tail <= 0)
  packed = torch.zeros((bsz, d, r + 2), dtype=torch.float32, device=dev)   # DS:555-592
  packed[:, :, :r] = eigvecs
  packed[:, :r, -2] = inverted
  packed[:, 0, -1] = new_const
  packed[:, 1, -1] = new_tail
  packed[:, d - r:, -1] = deflated
  packed[:, -1, -2] = has_zeros.to(torch.float32)
  ok = conv.cpu().tolist()
  return [(packed[j], _metrics(0.0)) if ok[j] else None for j in range(bsz)]


def _fd_update_root_batched(calls) -> list:
  """_fd_update_root for a list of keyword dicts.  The eigen-step of all of them runs
  batched: covariance updates of size >= SUBSPACE_MIN_N take their leading rank+1
  eigenpairs from subspace.top_eigenpairs_batched (MFMA GEMMs), the others (and any
  that does not converge) the full blocked-Jacobi eigendecomposition."""
  n_calls = len(calls)
  results = [None] * n_calls
  preps = [None] * n_calls
  # stacked fast path for groups of equal-sized factors without padding (config 5)
  by_key = {}
  for i, kw in enumerate(calls):
    key = _fd_group_key(kw)
    if key is not None:
      by_key.setdefault(key, []).append(i)
  for key, idxs in by_key.items():
    for i, res in zip(idxs, _fd_update_root_group([calls[i] for i in idxs], key)):
      results[i] = res
  for i, kw in enumerate(calls):
    if results[i] is not None:
      continue
    preps[i] = _fd_prepare(**{k: v for k, v in kw.items()
                              if k not in ("generate_training_metrics", "generate_fd_metrics")})
    preps[i]["want_fd"] = bool(kw.get("generate_fd_metrics") and
                               kw.get("generate_training_metrics", True))
    if preps[i]["want_fd"] and preps[i]["ps"] != 0:
      _fd_keep_for_diagnostics(preps[i], kw)
    if preps[i]["ps"] == 0:  # DS:1284-1288; the reference still returns an FDDiagnostics (zeros)
      m0 = _metrics(0.0)
      if preps[i]["want_fd"]:
        from .state import FDDiagnostics
        m0 = m0.replace(fd=FDDiagnostics())
      results[i] = (torch.zeros_like(kw["prev"]), m0)
  todo = [i for i in range(n_calls) if results[i] is None]
  # leading eigenpairs, batched by (size, rank)
  eig = {}
  groups = {}
  for i in todo:
    ps, rank = preps[i]["ps"], preps[i]["rank"]
    if ps >= SUBSPACE_MIN_N and 4 * (rank + 33) <= ps:
      groups.setdefault((ps, rank), []).append(i)
  full = [i for i in todo if not any(i in g for g in groups.values())]
  for (ps, rank), idxs in groups.items():
    from . import subspace
    e, v, conv, _ = subspace.top_eigenpairs_batched([preps[i]["c"] for i in idxs], rank + 1)
    conv = conv.cpu().tolist()
    for j, i in enumerate(idxs):
      if conv[j]:
        eig[i] = (e[j], v[j])
      else:
        full.append(i)
  if full:
    es, us = kernels.eigh_batched([preps[i]["c"] for i in full])  # ascending, like LAPACK
    for i, e, u in zip(full, es, us):
      eig[i] = (e.flip(0), u.flip(1))
  for i in todo:
    results[i] = _fd_finish(preps[i], *eig[i])
  return results


def _fd_prepare(new_grad, p, rank=0, ridge_epsilon=1e-6, error_tolerance=1e-6,
                relative_matrix_epsilon=True, decay=1.0, padding_start=None, prev=None,
                new_grad_is_gram=False):
  """Everything before the eigendecomposition: the unpadded covariance update C."""
  assert prev is not None and rank > 0
  max_size = new_grad.shape[0]
  assert list(new_grad.shape) == [max_size, max_size]
  pd = _precond_dim(rank, max_size)
  assert list(prev.shape) == [max_size, pd] and rank + 2 == pd and rank + 2 < max_size
  dev = new_grad.device
  ps = max_size if padding_start is None else int(padding_start)
  st = dict(ps=ps, rank=rank, p=p, decay=decay, max_size=max_size, dev=dev)
  if ps == 0:
    return st
  sketch_dr, fwd_eigvals_r, _, _, tail, _ = _fd_low_rank_unpack(prev, rank)
  max_ev = float(fwd_eigvals_r[0]) if relative_matrix_epsilon else 1.0
  ridge = ridge_epsilon * max(max_ev, error_tolerance)
  active_d = (torch.arange(max_size, device=dev) < ps).to(torch.float32)
  active_r = (torch.arange(rank, device=dev) < ps).to(torch.float32)
  sketch_dr = sketch_dr * active_d[:, None] * active_r
  fwd = (fwd_eigvals_r + ridge) * active_r
  weighted = (sketch_dr * torch.sqrt(fwd))[:ps].contiguous()          # [ps, r]

  # C = decay * W W^T + R R^T on the unpadded part (the SVD's u, s^2)
  if new_grad_is_gram:
    gram = new_grad[:ps, :ps].contiguous()
  else:
    rr = new_grad[:ps, :ps].contiguous()
    gram = kernels.matmul(rr, rr, transb=True)
  c = decay * kernels.matmul(weighted, weighted, transb=True) + gram
  st["c"] = 0.5 * (c + c.T)
  st["tail"] = tail
  st["weighted"] = weighted
  return st


def _fd_keep_for_diagnostics(st, kw):
  """What FDDiagnostics (DS:197-335) needs besides the decomposition."""
  st["new_grad"] = kw["new_grad"]
  st["new_grad_is_gram"] = bool(kw.get("new_grad_is_gram"))


def _fd_diagnostics(st, rho_t, new_tail, deflated, eigvecs, num_neg, num_zero_init,
                    num_unsafe, num_has_padding, top_sq):
  """FDDiagnostics.create (DS:262-335) + the fit errors of DS:1232-1241.
  The reference measures the fit on its SVD factors; here `updated updated^T = C`, so
  total_frob = ||updated||_F^2 = trace(C), and the top-k residual of an exact SVD is
  sum_{i >= k} s_i^2 = trace(C) - sum_{i < k} s_i^2 (square_frob and heuristic_frob, which the
  reference evaluates separately in float32, coincide).  Quantities that depend on WHICH
  square factor R of the Gram matrix was passed (entrywise_err, new_grad_abs_max and the
  two sparsities — R is not unique in the reference either, it is a QR factor) are computed
  from R when the caller passed one and are NaN when the slot holds the Gram matrix itself
  (`new_grad_is_gram`, this build's optimizer mode)."""
  from .state import FDDiagnostics
  ps, rank, max_size, dev = st["ps"], st["rank"], st["max_size"], st["dev"]
  f32 = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev)
  nan = f32(float("nan"))
  nz = deflated != 0
  eig_max = deflated.max()
  eig_min = torch.where(nz, deflated, eig_max).min()
  c = st["c"]
  total_frob = torch.diagonal(c).sum()
  resid = torch.clamp(total_frob - top_sq.sum(), min=0.0)
  g = st["new_grad"][:ps, :ps]
  if st["new_grad_is_gram"]:
    gram = g
    abs_max = sparsity = col_sparsity = entrywise = nan
  else:
    gram = kernels.matmul(g.contiguous(), g.contiguous(), transb=True)
    abs_max = g.abs().max()
    sparsity = (g == 0).sum().to(torch.float32) / float(ps * ps)
    col_sparsity = (g.abs().sum(dim=0) == 0).sum().to(torch.float32) / float(ps)
    # ||updated - U_k U_k^T updated||_1 / (ps^2 + ps rank), updated = [sqrt(decay) W | R]
    upd = torch.cat([st["weighted"] * (st["decay"] ** 0.5), g], dim=1).contiguous()
    uk = st["u_top"]
    proj = kernels.matmul(uk, kernels.matmul(uk, upd, transa=True))
    entrywise = (upd - proj).abs().sum() / float(ps * ps + ps * rank)
  _, ggt_max = kernels.power_iteration(gram.contiguous(), 100, 1e-6)
  cross = kernels.matmul(eigvecs.contiguous(), eigvecs.contiguous(), transa=True)
  ortho = (cross - torch.diag(torch.diagonal(cross))).abs().max()
  return FDDiagnostics(
      size_max_size=f32(max_size), size_rank=f32(rank), size_padding_start=f32(ps),
      rho=f32(rho_t), tail=f32(new_tail), eig_sparsity=(deflated == 0).to(torch.float32).mean(),
      eig_max=eig_max, eig_min=eig_min, new_grad_abs_max=abs_max, new_grad_sparsity=sparsity,
      new_grad_col_sparsity=col_sparsity, ggt_eig_max=ggt_max,
      ggt_intrinsic_dimension=torch.diagonal(gram).sum() / ggt_max, max_ortho_err=ortho,
      num_neg_eigs=f32(num_neg), num_zero_initial_eigs=f32(num_zero_init),
      num_unsafe_norms=f32(num_unsafe), num_has_padding=f32(num_has_padding),
      square_frob=resid, heuristic_frob=resid, entrywise_err=entrywise, total_frob=total_frob)


def _fd_finish(st, e, u):
  """Everything after it.  `e`: eigenvalues of C DESCENDING (all ps of them, or at least
  the leading rank+1), `u`: the matching eigenvectors in columns."""
  ps, rank, p, decay, max_size, dev = (st[k] for k in ("ps", "rank", "p", "decay",
                                                        "max_size", "dev"))
  tail = st["tail"]
  # Rank-deficient updates (a vector parameter's Gram has rank 1) give the
  # reference EXACT zero singular values, because it takes the SVD of a thin factor
  # (DS:1179-1193); they steer has_zeros / the deflation masks.  A float32
  # eigensolver returns +-n*eps*lambda_max there instead, so eigenvalues inside that
  # noise band are snapped to zero.
  noise = ps * 1.2e-7 * torch.clamp(e.max(), min=0.0)
  e = torch.where(e <= noise, torch.zeros_like(e), e)
  s = torch.sqrt(torch.clamp(e, min=0.0))
  # the padded problem has max_size singular values: pad with zeros
  s_full = torch.zeros((max(max_size, rank + 1),), dtype=torch.float32, device=dev)
  m = min(int(s.shape[0]), rank + 1)
  s_full[:m] = s[:m]
  cutoff = s_full[rank]
  rho_t = cutoff ** 2
  top_eigs = s_full[:rank]
  deflated = (top_eigs - cutoff) * (top_eigs + cutoff)
  eigvecs = torch.zeros((max_size, rank), dtype=torch.float32, device=dev)
  k = min(rank, ps, int(u.shape[1]))
  eigvecs[:ps, :k] = u[:, :k]
  tail = tail * decay
  new_tail = tail + rho_t
  alpha = -1.0 / p
  new_const = torch.where(new_tail <= 0, torch.zeros_like(new_tail), new_tail ** alpha)
  new_tail = torch.where(new_tail <= 0, torch.zeros_like(new_tail), new_tail)
  num_neg = (deflated < 0).sum()
  num_zero_init = (deflated == 0.0).sum()
  deflated = torch.where(deflated <= 0, torch.zeros_like(deflated), deflated)
  eigvecs = eigvecs * (deflated > 0)
  norms = torch.linalg.vector_norm(eigvecs, dim=0)
  safe = (0.99 <= norms) & (norms <= 1.01)
  num_unsafe = (~safe).sum() - (num_neg + num_zero_init)   # DS:1217-1219
  eigvecs = eigvecs * safe
  deflated = deflated * safe
  eigvecs = eigvecs / torch.where(safe, norms, torch.ones_like(norms))
  padding_ix = (torch.arange(max_size, device=dev) >= ps).to(torch.float32)
  padding_mass = (eigvecs * padding_ix[:, None]).abs().sum(dim=0)
  has_pad = (padding_mass > 0.01).to(torch.float32)
  eigvecs = eigvecs * (1 - has_pad)
  deflated = deflated * (1 - has_pad)
  upshifted = torch.square(top_eigs) + tail
  upshifted = upshifted * (deflated > 0.0)
  upshifted = torch.where(upshifted <= 0, torch.zeros_like(upshifted), upshifted)
  inverted = torch.where(upshifted <= 0, torch.zeros_like(upshifted), upshifted ** alpha)
  has_zeros = bool((deflated <= 0).any()) or bool((new_tail <= 0).any())
  packed = _fd_low_rank_pack(eigvecs, deflated, inverted, new_const, new_tail,
                             has_zeros, rank)
  metrics = _metrics(0.0)
  if st.get("want_fd"):
    st["u_top"] = u[:, :k].contiguous()
    metrics = metrics.replace(fd=_fd_diagnostics(
        st, rho_t, new_tail, deflated, eigvecs, num_neg, num_zero_init, num_unsafe,
        has_pad.sum(), torch.square(top_eigs)))
  return packed, metrics
