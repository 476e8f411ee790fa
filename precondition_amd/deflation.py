"""Top-k deflated inverse p-th root: the `lobpcg_topk_precondition` branch of
matrix_inverse_pth_root (DS:787-812, 813-817, 889-928).

Reference flow per block A (after padding masks):
  (theta, V)  <- top-k eigenpairs of A                  (jax LOBPCG, third-party)  DS:795-797
  A'          <- A - V diag(theta - min theta) V^T      (deflation)                DS:804-812
  max_ev      <- max theta  (no power iteration)                                   DS:813-817
  H'          <- coupled-Newton inverse p-th root of A' (ridge from max_ev)        DS:830-888
  H           <- H' - V diag(pth_diff) V^T,
                 pth_diff = (eps + min theta)^(-1/p) - (eps + theta)^(-1/p)        DS:891-902
  error       <- max |H^p (A + eps I) - I| of the UNconditioned problem            DS:910-928
Here the eigenpairs come from subspace.top_eigenpairs_batched (Chebyshev-filtered
subspace iteration on the MFMA GEMM, converged to 1e-5 residual — at least what LOBPCG
reaches in its iteration budget), the root from ps_newton_root_batched_maxev_f32 and the
rank-k updates / diagnostics from the grouped GEMM; O(k) scalars are torch ops.

Parity note: jax's lobpcg_standard is third-party and absent from /root/reference, so no
golden vectors exist for this branch.  It is pinned by the reference's own criterion
(DST:432-480: the preconditioned root's spectrum / entrywise error within 2x of the plain
root's) and by the float64 closed form.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .state import InversePthRootDiagnostics, LOBPCGDiagnostics


def _mat_power_grouped(K, mats: Sequence[torch.Tensor], p: int) -> List[torch.Tensor]:
  """M^p for a list of square matrices (binary powering, grouped GEMM launches)."""
  power: List[Optional[torch.Tensor]] = [None] * len(mats)
  cur = list(mats)
  e = int(p)
  while e > 0:
    if e & 1:
      nxt = [torch.empty_like(m) for m in cur]
      items = [(c, pw, o, False, False) for c, pw, o in zip(cur, power, nxt) if pw is not None]
      if items:
        K.gemm_grouped(items)
      power = [o if pw is not None else c for c, pw, o in zip(cur, power, nxt)]
    e >>= 1
    if e > 0:
      sq = [torch.empty_like(m) for m in cur]
      K.gemm_grouped([(c, c, o, False, False) for c, o in zip(cur, sq)])
      cur = sq
  return power  # p >= 1


def _root_diagnostics(K, roots, mats, p) -> List[InversePthRootDiagnostics]:
  """InversePthRootDiagnostics.create (DS:122-146) for a list of (root, matrix)."""
  pw = _mat_power_grouped(K, roots, p)
  prod = [torch.empty_like(m) for m in mats]
  K.gemm_grouped([(a, b, c, False, False) for a, b, c in zip(pw, mats, prod)])
  out = []
  for m in prod:
    n = m.shape[0]
    d = torch.diagonal(m)
    diag_err = (d - 1.0).abs()
    off = (m - torch.diag(d)).abs()
    out.append(InversePthRootDiagnostics(
        max_diag_error=diag_err.max(), avg_diag_error=diag_err.mean(),
        max_off_diag_error=off.max(),
        avg_off_diag_error=off.sum() / max(n * n - n, 1),
        p=torch.tensor(float(p), dtype=torch.float32, device=m.device)))
  return out


def matrix_inverse_pth_root_deflated_batched(
    matrices: Sequence[torch.Tensor], ps: Sequence[int],
    padding_starts: Optional[Sequence[int]] = None, topk: int = 0, max_iter: int = 0,
    num_iters: int = 100, ridge_epsilon: float = 1e-6, error_tolerance: float = 1e-6,
    relative_matrix_epsilon: bool = True, out: Optional[Sequence[torch.Tensor]] = None):
  """Returns (roots, metrics [batch, 8], diagnostics): diagnostics[b] is a dict with the
  TrainingMetrics fields lobpcg_diagnostics, conditioned_inverse_pth_root_diagnostics and
  inverse_pth_root_diagnostics (DS:915-928)."""
  from . import kernels as K, subspace
  from .distributed_shampoo import _EPSILON, _pth_root_difference
  # jax's lobpcg_standard (third party, absent here: parity unpinned) runs at most
  # `topk if lobpcg_max_iter == 0 else lobpcg_max_iter` iterations (DS:795-797).  The block
  # method below is a different iteration (Chebyshev-filtered subspace iteration, one outer
  # round = up to 12 filtered products + Rayleigh-Ritz); the same number is honoured as
  # the cap on its outer rounds, and `lobpcg_iters` reports the rounds it ran.
  outer_cap = max(1, min(50, int(topk) if not max_iter else int(max_iter)))
  batch = len(matrices)
  dev = matrices[0].device
  k = int(topk)
  sizes = [int(m.shape[0]) for m in matrices]
  pads = list(sizes) if padding_starts is None else [min(int(x), n) for x, n in
                                                     zip(padding_starts, sizes)]
  if out is None:
    out = [torch.empty((n, n), dtype=torch.float32, device=dev) for n in sizes]
  metrics = torch.zeros((batch, 8), dtype=torch.float32, device=dev)
  diags = [dict(lobpcg_diagnostics=LOBPCGDiagnostics(),
                conditioned_inverse_pth_root_diagnostics=InversePthRootDiagnostics(),
                inverse_pth_root_diagnostics=InversePthRootDiagnostics())
           for _ in range(batch)]
  live = [b for b in range(batch) if pads[b] > 0]
  for b in range(batch):
    if pads[b] == 0:  # all padding: zeros, error 0 (DS:930-937)
      out[b].zero_()
      metrics[b, 2], metrics[b, 4] = 1.0, 1.0
    elif pads[b] <= k:
      raise ValueError(f"lobpcg_topk_precondition={k} needs blocks larger than k "
                       f"(block {b} has {pads[b]} rows)")
  if not live:
    return list(out), metrics, diags

  # the unpadded problems (padding rows / columns are masked out, DS:777-784)
  a_eff = [matrices[b][:pads[b], :pads[b]].contiguous() for b in live]
  theta = [None] * len(live)
  vecs = [None] * len(live)
  iters = [0] * len(live)
  groups = {}
  for j, b in enumerate(live):
    groups.setdefault(pads[b], []).append(j)
  for n, js in groups.items():
    e, v, conv, info = subspace.top_eigenpairs_batched([a_eff[j] for j in js], k,
                                                       max_outer=outer_cap,
                                                       oversample=max(8, min(31, n - k)))
    for t, j in enumerate(js):
      theta[j], vecs[j], iters[j] = e[t], v[t], info["outer_iterations"]

  # deflate (DS:804-812) and root the conditioned problems with max_ev given
  deflated, max_ev = [], []
  items, wbufs = [], []
  for j in range(len(live)):
    th, v = theta[j], vecs[j]
    w = (v * torch.sqrt((th - th.min()).clamp_min(0.0))).contiguous()
    d = torch.empty_like(a_eff[j])
    items.append((w, w, d, False, True))
    wbufs.append(d)
    max_ev.append(th.max())
  K.gemm_grouped(items)
  deflated = [a - d for a, d in zip(a_eff, wbufs)]
  max_ev_t = torch.stack(max_ev) if relative_matrix_epsilon else None
  p_live = [int(ps[b]) for b in live]
  cond_roots, m = K.matrix_inverse_pth_root_batched(
      deflated, p_live, None, num_iters=num_iters, ridge_epsilon=ridge_epsilon,
      error_tolerance=error_tolerance, relative_matrix_epsilon=relative_matrix_epsilon,
      max_ev=max_ev_t)

  # re-deflate (DS:891-902) and the diagnostics of both problems (DS:909-928)
  roots_eff, cond_damped, uncond_damped = [], [], []
  items, upd = [], []
  ridge = []
  for j in range(len(live)):
    th, v, p = theta[j], vecs[j], p_live[j]
    mev = max_ev[j] if relative_matrix_epsilon else torch.ones((), device=dev)
    r = ridge_epsilon * torch.clamp(mev, min=_EPSILON)
    ridge.append(r)
    diff = _pth_root_difference(r, th.min(), th, p)
    w = (v * torch.sqrt(diff.clamp_min(0.0))).contiguous()
    d = torch.empty_like(a_eff[j])
    items.append((w, w, d, False, True))
    upd.append(d)
  K.gemm_grouped(items)
  for j in range(len(live)):
    n = a_eff[j].shape[0]
    eye = torch.eye(n, dtype=torch.float32, device=dev)
    roots_eff.append((cond_roots[j] - upd[j]).contiguous())
    retries = m[j, 4]
    cond_damped.append(deflated[j] + (ridge[j] * torch.pow(10.0, retries)) * eye)
    uncond_damped.append(a_eff[j] + ridge[j] * eye)
  by_p = {}
  for j, p in enumerate(p_live):
    by_p.setdefault(p, []).append(j)
  cd, ud = [None] * len(live), [None] * len(live)
  for p, js in by_p.items():
    for j, d in zip(js, _root_diagnostics(K, [cond_roots[j] for j in js],
                                          [cond_damped[j] for j in js], p)):
      cd[j] = d
    for j, d in zip(js, _root_diagnostics(K, [roots_eff[j] for j in js],
                                          [uncond_damped[j] for j in js], p)):
      ud[j] = d

  # LOBPCGDiagnostics.create (DS:171-194)
  av = [torch.empty_like(v) for v in vecs]
  K.gemm_grouped([(a, v.contiguous(), o, False, False) for a, v, o in zip(a_eff, vecs, av)])
  for j, b in enumerate(live):
    th, v = theta[j], vecs[j]
    unnorm = torch.linalg.vector_norm(av[j] - th * v, dim=0)
    cons = unnorm / (torch.linalg.vector_norm(av[j], dim=0) + th)
    gram = K.matmul(v.contiguous(), v.contiguous(), transa=True)  # k x k
    ortho = gram - torch.diag(torch.diagonal(gram))
    lob = LOBPCGDiagnostics(
        lobpcg_iters=torch.tensor(float(iters[j]), device=dev),
        max_consistency_error=cons.max(), avg_consistency_error=cons.mean(),
        avg_orthogonality_error=ortho.sum() / max(k * (k - 1), 1),
        max_eigenvalue=th.max(), min_eigenvalue=th.min(),
        num_topk_eigenvectors=torch.tensor(float(k), device=dev))
    n_eff, n_full = pads[b], sizes[b]
    o = out[b]
    if n_eff < n_full:
      o.zero_()
    o[:n_eff, :n_eff] = roots_eff[j]
    metrics[b] = m[j]
    # the reported error is the UNconditioned problem's (DS:909-922)
    metrics[b, 0] = torch.maximum(ud[j].max_diag_error, ud[j].max_off_diag_error)
    diags[b] = dict(lobpcg_diagnostics=lob, conditioned_inverse_pth_root_diagnostics=cd[j],
                    inverse_pth_root_diagnostics=ud[j])
  return list(out), metrics, diags
