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
                                    relative_matrix_epsilon=True, eigh=False, out=None,
                                    options=None):
  del options  # execution modes of the HIP library (ps_options): nothing to select on the oracle
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


def matmul(a, b, transa=False, transb=False):
  x = a.cpu().numpy().T if transa else a.cpu().numpy()
  y = b.cpu().numpy().T if transb else b.cpu().numpy()
  return torch.from_numpy((x @ y).astype(np.float32))


class _TM:

  def __init__(self, err):
    self.inverse_pth_root_errors = torch.tensor(err, dtype=torch.float32)


def low_rank_root(matrix, p, compression_rank=0, ridge_epsilon=1e-6, error_tolerance=1e-6,
                  relative_matrix_epsilon=True, padding_start=None, prev=None):
  val, err = orc.low_rank_root(matrix.cpu().numpy(), p, compression_rank, ridge_epsilon,
                               error_tolerance, relative_matrix_epsilon, padding_start)
  return torch.from_numpy(val), _TM(err)


def fd_update_root(new_grad, p, rank=0, ridge_epsilon=1e-6, error_tolerance=1e-6,
                   relative_matrix_epsilon=True, decay=1.0, padding_start=None, prev=None,
                   new_grad_is_gram=False, **_):
  g = new_grad.cpu().numpy()
  if new_grad_is_gram:  # any factor with R R^T = Gram serves (DS:1179-1193)
    w, v = np.linalg.eigh(g.astype(np.float64))
    w = np.where(w <= g.shape[0] * 1.2e-7 * max(w.max(), 0.0), 0.0, w)  # keep exact rank deficiency
    # (the reference's QR factor of a thin matrix has exact zero columns)
    # nonzero columns first, like the triangular factor (the reference masks the
    # factor's COLUMNS >= padding_start, DS:1173-1174)
    g = (v[:, ::-1] * np.sqrt(w[::-1])).astype(np.float32)
  val = orc.fd_update_root(g, p, rank, ridge_epsilon, error_tolerance,
                           relative_matrix_epsilon, decay, padding_start, prev.cpu().numpy())
  return torch.from_numpy(val), _TM(0.0)


def fd_update_root_batched(calls):
  return [fd_update_root(**kw) for kw in calls]


def gemm_grouped(items):
  for a, b, c, ta, tb in items:
    c.copy_(matmul(a, b, transa=ta, transb=tb))


_NP_Q = {torch.int8: np.int8, torch.int16: np.int16}


def quantize_grouped(fvalues, quantized_dtype, extract_diagonal=False, out=None):
  from oracle import quantization_oracle as qorc
  res = []
  for i, f in enumerate(fvalues):
    q, d, b = qorc.quantize(f.cpu().numpy(), _NP_Q[quantized_dtype], extract_diagonal)
    t = (torch.from_numpy(q), torch.from_numpy(d) if extract_diagonal else [],
         torch.from_numpy(np.asarray(b, dtype=np.float32)))
    if out is not None:
      out[i][0].copy_(t[0].view(out[i][0].shape))
      if extract_diagonal:
        out[i][1].copy_(t[1])
      out[i][2].copy_(t[2].view(out[i][2].shape))
      t = out[i]
    res.append(t)
  return res


def dequantize_grouped(items, out=None):
  from oracle import quantization_oracle as qorc
  res = []
  for i, (q, d, b) in enumerate(items):
    extract = not (isinstance(d, list) and not d)
    f = qorc.to_float(q.cpu().numpy(), d.cpu().numpy() if extract else [], b.cpu().numpy(),
                      q.cpu().numpy().dtype, extract)
    t = torch.from_numpy(np.asarray(f, dtype=np.float32))
    if out is not None:
      out[i].copy_(t)
      t = out[i]
    res.append(t)
  return res
