"""Optimizer states across the two implementations (Frequent-Directions mode).

Every state type mirrors the reference's (names, field order: state.py), so a checkpoint written by the
reference maps leaf for leaf onto a ShampooState of this build -- with ONE difference of content: in
Frequent-Directions mode the reference keeps, in the statistics slot of a compressed factor, the zero-padded
triangular factor R^T of the gradient block (qr(x^T, mode='r'), LAPACK sgeqrf signs, DS:1497-1505), this build
the Gram matrix R R^T = x x^T itself.  Only R R^T is ever consumed (_fd_update_root stacks [sqrt(beta) W | R]
and takes its left singular pairs, DS:1179-1193), and producing R on the GPU would cost a Householder QR per
block and step (2 k d^2 flops, more than the whole sketch update) for a quantity nothing reads.  The slot is
also overwritten by the next statistics step before any root uses it (FD requires statistics_compute_steps ==
preconditioning_compute_steps, DS:2035-2040).  So the states are interchangeable through a conversion of that
one slot:

  import_reference_state(state, compression_rank)   R -> R R^T           (a reference checkpoint continues here)
  export_reference_state(state, compression_rank)   Gram -> a factor F, F F^T = Gram (continues in the reference;
                                                    F = V sqrt(Lambda), not triangular: any factor is equivalent)

tests/test_interop.py continues an interrupted run of the reference's own source (tests/golden/fd_resume.npz)
from its converted state and compares the following updates."""
from __future__ import annotations

import torch

from . import pytree
from .blocking import _should_compress
from .state import ParameterStats, ShampooState


def _gram_of_factor(r: torch.Tensor) -> torch.Tensor:
  r = r.to(torch.float32).contiguous()
  if r.is_cuda:
    from . import kernels
    return kernels.matmul(r, r, transb=True)
  return r @ r.T


def _factor_of_gram(g: torch.Tensor) -> torch.Tensor:
  g = g.to(torch.float32).contiguous()
  if g.is_cuda:
    from . import kernels
    es, vs = kernels.eigh_batched([g])
    e, v = es[0], vs[0]
  else:
    e, v = torch.linalg.eigh(g.double())
    e, v = e.float(), v.float()
  return (v * torch.sqrt(torch.clamp(e, min=0.0))).contiguous()


def _convert(state: ShampooState, compression_rank: int, fn) -> ShampooState:
  if compression_rank <= 0:
    return state
  flat, treedef = pytree.tree_flatten(state.stats, is_leaf=lambda x: isinstance(x, ParameterStats))
  out = []
  for st in flat:
    if not isinstance(st, ParameterStats) or not len(st.statistics):
      out.append(st)
      continue
    stats = [fn(s) if (isinstance(s, torch.Tensor) and s.dim() == 2 and s.shape[0] == s.shape[1] and
                       _should_compress(compression_rank, int(s.shape[0]))) else s
             for s in st.statistics]
    out.append(st._replace(statistics=stats))
  return ShampooState(count=state.count, stats=treedef.unflatten(out))


def import_reference_state(state: ShampooState, compression_rank: int) -> ShampooState:
  """A state whose leaves hold the REFERENCE's values (frequent_directions=True, rank `compression_rank`):
  the statistics slots of compressed factors (triangular factors R) become Gram matrices R R^T."""
  return _convert(state, compression_rank, _gram_of_factor)


def export_reference_state(state: ShampooState, compression_rank: int) -> ShampooState:
  """The inverse direction: Gram matrices become factors F with F F^T = Gram."""
  return _convert(state, compression_rank, _factor_of_gram)
