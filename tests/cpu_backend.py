"""CPU stand-in for precondition_amd.kernels, for TESTS of the host logic only
(tree bookkeeping, grafting/momentum, gloo sharding).  It routes the three
numerical entry points through the oracle; the product never imports this."""
import numpy as np
import torch

from oracle import shampoo_oracle as orc


def stats_update_grouped(items, w1, w2):
  for g, axis, sin, sout in items:
    new = orc.gram_weighted_update(sin.cpu().numpy(), g.cpu().numpy(), axis, w1, w2)
    sout.copy_(torch.from_numpy(new))


def matrix_inverse_pth_root_batched(matrices, ps, padding_starts=None, num_iters=100,
                                    ridge_epsilon=1e-6, error_tolerance=1e-6,
                                    relative_matrix_epsilon=True, eigh=False, out=None):
  roots, rows = [], []
  for i, (m, p) in enumerate(zip(matrices, ps)):
    pad = None if padding_starts is None else int(padding_starts[i])
    if eigh:
      h, met = orc.matrix_inverse_pth_root_eigh(
          m.cpu().numpy(), p, ridge_epsilon=ridge_epsilon,
          relative_matrix_epsilon=relative_matrix_epsilon, padding_start=pad)
    else:
      h, met = orc.matrix_inverse_pth_root(
          m.cpu().numpy(), p, num_iters=num_iters, ridge_epsilon=ridge_epsilon,
          error_tolerance=error_tolerance,
          relative_matrix_epsilon=relative_matrix_epsilon, padding_start=pad)
    t = torch.from_numpy(h)
    if out is not None:
      out[i].copy_(t)
      t = out[i]
    roots.append(t)
    rows.append([met["inverse_pth_root_errors"], met["inverse_pth_root_iters"],
                 met["final_error_ratio"], met["max_eigen_value"],
                 met["total_retries"], met["inverse_pth_root_iters"], 0.0, 0.0])
  return roots, torch.tensor(rows, dtype=torch.float32)


def tensordot_axis0(g, pc):
  return torch.from_numpy(
      np.tensordot(g.cpu().numpy(), pc.cpu().numpy(), axes=[[0], [0]]).astype(np.float32))
