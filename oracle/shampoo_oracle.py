"""CPU oracle for the Shampoo preconditioner-compute hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain NumPy float32 *restatement* of the reference algorithm
(google-research/precondition, ``precondition/distributed_shampoo.py``, cited
below as ``DS:<line>``).  It exists so that the HIP kernels can be checked
against something that follows the reference line by line.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package (``precondition_amd``) never does and fails loudly if
its HIP library is missing.

Parity pin: ``tools/gen_golden.py`` runs the reference's own source (imported
from /root/reference over a NumPy stand-in for jax) on seeded inputs and
stores inputs+outputs under ``tests/golden/``; at generation time this oracle
is asserted to agree **bit for bit** with those outputs on the generating
machine (same NumPy/OpenBLAS op sequence), and ``tests/test_oracle_golden.py``
re-checks it against the committed vectors to a tight tolerance on any machine
(OpenBLAS picks different kernels per CPU, so bits may differ elsewhere).
Third-party arithmetic the reference delegates to jax/XLA is *unpinned* upstream
(pyproject.toml:17-28 lists a bare "jax"); here dot is NumPy/OpenBLAS float32 and
eigh / svd / qr are the SINGLE-PRECISION LAPACK routines jax's CPU path runs (ssyevd,
sgesdd, sgeqrf through scipy.linalg.lapack: oracle/lapack32.py).  ``lapack="f64"``
selects NumPy's float64-internal versions instead: an accuracy yardstick, not the
parity target (NumPy computes float32 inputs in double — rounds 1-5 used it by mistake).

Everything is float32 because ``_MAT_INV_PTH_ROOT_DTYPE = jnp.float64``
(DS:38) silently degrades to float32 unless jax_enable_x64 is set (DS:35-38),
which neither the library nor its tests do.
"""
from __future__ import annotations

import numpy as np

from oracle import lapack32 as _lapack

F32 = np.float32


def _lapack_fns(lapack):
  """The LAPACK-delegated routines (DS:1007, 1071, 1193, 1502): ``"f32"`` = the single-precision
  routines the reference's JAX CPU path runs (ssyevd / sgesdd / sgeqrf: the PARITY target);
  ``"f64"`` = NumPy's float64-internal versions (accuracy yardstick only; oracle/lapack32.py)."""
  if lapack == "f32":
    return _lapack.eigh32, _lapack.svd32, _lapack.qr_r32
  assert lapack == "f64", lapack
  return _lapack.eigh64, _lapack.svd64, _lapack.qr_r64
_EPSILON = 1e-25  # DS:41


# ---------------------------------------------------------------------------
# DS:595-652 power_iteration
# ---------------------------------------------------------------------------
def power_iteration(matrix, num_iters=100, error_tolerance=1e-6,
                    padding_start=None):
  """Returns (v, s, iters_run).  Follows DS:625-652 statement by statement."""
  matrix = np.asarray(matrix, dtype=F32)
  n = matrix.shape[-1]
  # DS:642-646: fixed seed; note uniform(size=n) is a prefix of uniform(size=m>n).
  v_0 = np.random.RandomState(1729).uniform(-1.0, 1.0, n).astype(F32)
  if padding_start is not None:
    v_0 = v_0 * (np.arange(n, dtype=np.int32) < padding_start).astype(F32)
  i = 0
  new_v = v_0
  s = F32(0.0)
  run_step = True
  while i < num_iters and run_step:  # DS:627-629
    new_v = new_v / np.linalg.norm(new_v)  # DS:634
    s_v = np.einsum("ij,j->i", matrix, new_v)  # DS:636
    s_new = np.einsum("i,i->", new_v, s_v)  # DS:637
    run_step = bool(np.abs(s_new - s) > error_tolerance)  # DS:639
    i, new_v, s = i + 1, s_v, F32(s_new)
  v_out = new_v / np.linalg.norm(new_v)  # DS:651
  return v_out, s, i


# ---------------------------------------------------------------------------
# DS:655-678 mat_power
# ---------------------------------------------------------------------------
def mat_power(mat_m, p):
  """M^p by binary powering with the reference's multiplication order.

  The reference multiplies ``mat @ power`` (DS:671) starting from ``power=I``
  and squares ``mat`` once more than needed (DS:674); ``X @ I`` is exact in
  float32 and the trailing square is unused, so both are skipped here without
  changing a single bit of the result.
  """
  mat = np.asarray(mat_m, dtype=F32)
  power = None
  i = int(p)
  while i > 0:
    if i % 2 == 1:
      power = mat if power is None else np.matmul(mat, power)
    i //= 2
    if i > 0:
      mat = np.matmul(mat, mat)
  if power is None:  # p == 0
    power = np.eye(mat.shape[0], dtype=F32)
  return power


def mat_power_products(p):
  """Number of n^3 products mat_power *needs* for exponent p."""
  p = int(p)
  return int(np.floor(np.log2(p))) + bin(p).count("1") - 1 if p > 0 else 0


def newton_products_per_iter(p):
  """c(p) of SURVEY §8d: mat_power products + M update + H update."""
  return mat_power_products(p) + 2


# ---------------------------------------------------------------------------
# DS:702-940 matrix_inverse_pth_root (coupled Newton branch)
# ---------------------------------------------------------------------------
def matrix_inverse_pth_root(matrix, p, num_iters=100, ridge_epsilon=1e-6,
                            error_tolerance=1e-6, relative_matrix_epsilon=True,
                            padding_start=None):
  """Returns (root float32 [n,n], metrics dict of python floats)."""
  matrix = np.array(matrix, dtype=F32)  # DS:773
  n = matrix.shape[0]
  assert matrix.shape == (n, n)
  p = int(p)
  alpha = F32(-1.0 / p)  # DS:774
  identity = np.eye(n, dtype=F32)  # DS:775
  if padding_start is not None:  # DS:777-783
    ix = (np.arange(n, dtype=np.int32) < padding_start).astype(F32)
    matrix = matrix * ix[np.newaxis, :]
    matrix = matrix * ix[:, np.newaxis]
    identity = identity * ix
  if relative_matrix_epsilon:  # DS:814-828
    _, max_ev, _ = power_iteration(matrix, 100, 1e-6, padding_start)
  else:
    max_ev = F32(1.0)
  ridge = F32(F32(ridge_epsilon) * np.maximum(F32(max_ev), F32(_EPSILON)))  # DS:830
  max_error_ratio = F32(1.2)  # DS:834
  one_minus_alpha = F32(1) - alpha

  # DS:860 init_outer_state = (0, identity, 1000.0, 100, 1.0, True)
  tries = 0
  resultant = identity
  error = F32(1000.0)
  iters = 100
  error_ratio = F32(1.0)
  failed = True
  with np.errstate(all="ignore"):
    while failed and tries < 6:  # DS:862-864
      damped = matrix + (ridge * F32(10**tries)) * identity  # DS:869
      z = F32(1 + p) / (F32(2) * np.linalg.norm(damped))  # DS:870
      mat_m = damped * z  # DS:871
      err = np.max(np.abs(mat_m - identity))  # DS:872
      mat_h = identity * np.power(z, F32(1.0 / p))  # DS:873
      old_mat_h = mat_h
      it = 0
      ratio = F32(1.0)
      # DS:836-848
      while it < num_iters and err > error_tolerance and ratio < max_error_ratio:
        mat_m_i = one_minus_alpha * identity + alpha * mat_m  # DS:844
        new_mat_m = np.matmul(mat_power(mat_m_i, p), mat_m)  # DS:845
        new_mat_h = np.matmul(mat_h, mat_m_i)  # DS:846
        new_err = np.max(np.abs(new_mat_m - identity))  # DS:847
        ratio = F32(new_err / err)
        it, mat_m, old_mat_h, mat_h, err = it + 1, new_mat_m, mat_h, new_mat_h, new_err
      error = F32(np.max(np.abs(mat_m - identity)))  # DS:878
      is_converged = F32(1.0) if ratio < max_error_ratio else F32(0.0)  # DS:879
      resultant = is_converged * mat_h + (F32(1) - is_converged) * old_mat_h  # DS:880
      tries, iters, error_ratio = tries + 1, it, ratio
      failed = bool(error > 0.05)  # DS:858,882 (NaN > 0.05 is False)
  if padding_start is not None and padding_start == 0:  # DS:930-937
    resultant = np.zeros_like(resultant)
    error = F32(0.0)
  metrics = dict(
      inverse_pth_root_errors=float(error),
      inverse_pth_root_iters=float(iters),
      final_error_ratio=float(error_ratio),
      max_eigen_value=float(max_ev),
      total_retries=float(tries),
  )
  return resultant.astype(F32), metrics


# ---------------------------------------------------------------------------
# DS:943-1030 matrix_inverse_pth_root_eigh
# ---------------------------------------------------------------------------
def matrix_inverse_pth_root_eigh(matrix, p, ridge_epsilon=1e-6,
                                 error_tolerance=1e-6,
                                 relative_matrix_epsilon=True,
                                 padding_start=None, lapack="f32"):
  eigh_fn = _lapack_fns(lapack)[0]
  matrix = np.array(matrix, dtype=F32)
  n = matrix.shape[0]
  alpha = F32(-1.0 / int(p))
  identity = np.eye(n, dtype=F32)
  ix = None
  if padding_start is not None:  # DS:989-994
    ix = (np.arange(n, dtype=np.int32) < padding_start).astype(F32)
    matrix = matrix * ix[np.newaxis, :]
    matrix = matrix * ix[:, np.newaxis]
    identity = identity * ix
  if relative_matrix_epsilon:  # DS:995-1001 (note: tolerance = error_tolerance)
    _, max_ev, _ = power_iteration(matrix, 100, error_tolerance, padding_start)
  else:
    max_ev = F32(1.0)
  ridge = F32(F32(ridge_epsilon) * np.maximum(F32(max_ev), F32(error_tolerance)))  # DS:1005
  regularized = matrix + ridge * identity  # DS:1006
  if np.all(np.isfinite(regularized)):
    e, u = eigh_fn(regularized)  # DS:1007 (LAPACK ssyevd in float32, ascending)
  else:  # all-padding block: max_ev is 0/0; XLA yields NaNs where LAPACK raises
    e, u = np.full(n, np.nan, F32), np.full((n, n), np.nan, F32)
  e = e.astype(F32)
  u = u.astype(F32)
  if ix is not None:
    e = e * ix[::-1]  # DS:1010
  with np.errstate(all="ignore"):
    inv_e = np.where(e == 0.0, F32(0.0),
                     np.power(np.maximum(e, ridge), alpha)).astype(F32)  # DS:1012
  root = u * np.sqrt(inv_e)  # DS:1015
  # separate buffers force plain sgemm (NumPy would otherwise pick ssyrk for
  # X @ X.T, which rounds differently from a general dot)
  val = np.matmul(root, np.array(root.T))  # DS:1016
  recovered_e = np.matmul(np.array(u.T),
                          np.matmul(regularized, u))  # DS:1017
  eig_error = recovered_e - np.diag(e)  # DS:1018
  if ix is not None:
    eig_error = eig_error * ix[::-1]  # DS:1020
  error = F32(np.max(np.abs(eig_error)))  # DS:1021
  if padding_start is not None and padding_start == 0:  # DS:1024-1028
    val = np.zeros_like(val)
    error = F32(0.0)
  metrics = dict(
      inverse_pth_root_errors=float(error),
      inverse_pth_root_iters=0.0,
      final_error_ratio=0.0,
      max_eigen_value=0.0,  # DS:1022: only the error field is populated
      total_retries=0.0,
  )
  return val.astype(F32), metrics


# ---------------------------------------------------------------------------
# DS:1440-1470 gram_weighted_update ; DS:2635-2636 weights
# ---------------------------------------------------------------------------
# accuracy yardstick (no counterpart in the reference)
# ---------------------------------------------------------------------------
def eigh_root_float64(matrix, p, ridge_epsilon=1e-6, error_tolerance=1e-6,
                      relative_matrix_epsilon=True, padding_start=None):
  """Float64 closed form of the eigh root of the float32 matrix every solver is handed:
  D = A + ridge I is formed in float32 exactly as DS:995-1006 does (power-iteration lambda_max,
  float32 addition); the eigendecomposition and DS:1012-1016 are then exact to float64.  The
  distance of a float32 solver's root from this is ITS OWN rounding error (the reference's
  ssyevd, the build's solvers), not the shared rounding of the ridge addition."""
  a = np.asarray(matrix, F32)
  n = a.shape[0]
  ps = n if padding_start is None else int(padding_start)
  out = np.zeros((n, n), np.float64)
  if ps == 0:
    return out
  am = a[:ps, :ps]
  if relative_matrix_epsilon:
    _, max_ev, _ = power_iteration(am, 100, error_tolerance, None)
  else:
    max_ev = F32(1.0)
  ridge = F32(F32(ridge_epsilon) * np.maximum(F32(max_ev), F32(error_tolerance)))
  d32 = (am + ridge * np.eye(ps, dtype=F32)).astype(F32)
  w, v = np.linalg.eigh(d32.astype(np.float64))
  out[:ps, :ps] = (v * np.maximum(w, float(ridge)) ** (-1.0 / int(p))) @ v.T
  return out


# ---------------------------------------------------------------------------
def gram_weighted_update(old_stats, g, axis, w1, w2):
  g = np.asarray(g, dtype=F32)
  axes = [i for i in range(g.ndim) if i != axis]
  # second operand is a separate buffer so NumPy uses sgemm, not ssyrk
  gram = np.tensordot(g, np.array(g), axes=(axes, axes))  # DS:1469
  return (F32(w1) * np.asarray(old_stats, F32) + F32(w2) * gram).astype(F32)  # DS:1470


def stats_weights(beta2):
  """DS:2635-2636: w1 = beta2, w2 = 1 if beta2 == 1 else 1 - beta2."""
  w1 = beta2
  w2 = beta2 if beta2 == 1.0 else 1.0 - beta2
  return w1, w2


# ---------------------------------------------------------------------------
# DS:1324-1350 pad_square_matrix ; DS:1827-1846 batch/unbatch ;
# DS:2816-3010 _pmap_compute_preconditioners (numerical part only)
# ---------------------------------------------------------------------------
def pad_square_matrix(mat, max_size):
  mat = np.asarray(mat)
  rows, cols = mat.shape
  if rows != cols:
    raise ValueError("Must have rows == cols, instead got "
                     f"rows={rows}, cols={cols}")
  if cols > max_size:
    raise ValueError("Must have cols <= max_size. Instead got "
                     f"cols={cols}, max_size={max_size}.")
  if rows == max_size:
    return mat
  out = np.zeros((max_size, max_size), dtype=mat.dtype)
  out[:rows, :rows] = mat
  idx = np.arange(rows, max_size)
  out[idx, idx] = 1
  return out


def compute_preconditioners_reference_order(statistics, exponents,
                                            prev_preconditioners, num_devices,
                                            ridge_epsilon=1e-6,
                                            relative_matrix_epsilon=True,
                                            inverse_failure_threshold=0.1,
                                            eigh=False):
  """Emulates DS:2816-2950 for ``num_devices`` replicas on one host.

  Every statistic is padded to ``max_size`` (DS:2841-2843), the list is padded
  with identities / exponent 1 / padding_start 0 to a multiple of
  ``num_devices`` (DS:2844-2850), replica r computes the contiguous chunk
  ``[r*b, (r+1)*b)`` (DS:2862-2873, ``batch``), results are concatenated in
  replica order (``all_gather`` + ``unbatch``, DS:2876-2879), cropped to the
  original shape and selected against the previous preconditioner on NaN /
  error >= threshold (DS:2936-2950).

  Returns (new_preconditioners, metrics_list (incl. the padding entries),
  owner_rank per padded index).
  """
  num_statistics = len(statistics)
  if num_statistics == 0:
    return [], [], []
  max_size = max(s.shape[0] for s in statistics)
  packed = [pad_square_matrix(np.asarray(s, F32), max_size) for s in statistics]
  to_pad = -num_statistics % num_devices
  packed += [np.eye(max_size, dtype=F32) for _ in range(to_pad)]
  exps = list(exponents) + [1] * to_pad
  paddings = [s.shape[0] for s in statistics] + [0] * to_pad
  n_total = len(packed)
  b = n_total // num_devices
  roots, metrics, owners = [None] * n_total, [None] * n_total, [None] * n_total
  fn = matrix_inverse_pth_root_eigh if eigh else matrix_inverse_pth_root
  for r in range(num_devices):
    for j in range(r * b, (r + 1) * b):
      roots[j], metrics[j] = fn(
          packed[j], exps[j], ridge_epsilon=ridge_epsilon,
          relative_matrix_epsilon=relative_matrix_epsilon,
          padding_start=paddings[j])
      owners[j] = r
  new_p = []
  for j in range(num_statistics):
    err = metrics[j]["inverse_pth_root_errors"]
    n = statistics[j].shape[0]
    if np.isnan(err) or err >= inverse_failure_threshold:
      new_p.append(np.asarray(prev_preconditioners[j], F32))
    else:
      new_p.append(roots[j][:n, :n])
  return new_p, metrics, owners


# ---------------------------------------------------------------------------
# CPU-baseline helper for bench.py: the reference's *executed* op sequence.
# ---------------------------------------------------------------------------
def newton_root_reference_opcount(matrix, p, ridge_epsilon=1e-6,
                                  padding_start=None):
  """Same result as matrix_inverse_pth_root but executes the reference's
  redundant products too (the ``@ I`` and the unused trailing square of
  DS:670-674), so that its wall time is the reference's CPU op sequence.
  Used only as bench.py's ``cpu_baseline`` (kind "port")."""
  eye_cache = {}

  def ref_mat_power(mat, p):
    n = mat.shape[0]
    if n not in eye_cache:
      eye_cache[n] = np.eye(n, dtype=F32)
    power = eye_cache[n]
    i = int(p)
    while i > 0:
      if i % 2 == 1:
        power = np.matmul(mat, power)
      i //= 2
      mat = np.matmul(mat, mat)
    return power

  global mat_power
  saved = mat_power
  mat_power = ref_mat_power
  try:
    return matrix_inverse_pth_root(matrix, p, ridge_epsilon=ridge_epsilon,
                                   padding_start=padding_start)
  finally:
    mat_power = saved


# ---------------------------------------------------------------------------
# Low-rank / Frequent-Directions branch (BASELINE config 5)
# DS:520-537 _precond_dim/_should_compress ; DS:555-592 pack/unpack ;
# DS:1033-1120 _low_rank_root ; DS:1123-1290 _fd_update_root ; DS:1473-1505
# ---------------------------------------------------------------------------
def precond_dim(compression_rank, dim):
  if not compression_rank:
    return dim
  c = abs(compression_rank) + 2
  return dim if c >= dim else c


def fd_low_rank_unpack(pc, rank):
  r = abs(rank)
  pc = np.asarray(pc, F32)
  return (pc[:, :r], pc[-r:, -1], pc[:r, -2], pc[0, -1], pc[1, -1], bool(pc[-1, -2]))


def fd_low_rank_pack(eigvecs, deflated, inverted, const, tail, has_zeros, rank):
  r = abs(rank)
  d = eigvecs.shape[0]
  pc = np.zeros((d, r + 2), F32)
  pc[:, :r] = eigvecs
  pc[:r, -2] = inverted
  pc[0, -1] = const
  pc[1, -1] = tail
  pc[-r:, -1] = deflated
  pc[-1, -2] = F32(1.0 if has_zeros else 0.0)
  return pc


def low_rank_root(matrix, p, compression_rank, ridge_epsilon=1e-6, error_tolerance=1e-6,
                  relative_matrix_epsilon=True, padding_start=None, lapack="f32"):
  """DS:1033-1120, statement by statement (LAPACK ssyevd for jnp.linalg.eigh)."""
  eigh_fn = _lapack_fns(lapack)[0]
  matrix = np.array(matrix, F32)
  d = matrix.shape[0]
  alpha = F32(-1.0 / int(p))
  identity = np.eye(d, dtype=F32)
  ix = None
  if padding_start is not None:
    ix = (np.arange(d, dtype=np.int32) < padding_start).astype(F32)
    matrix = matrix * ix[np.newaxis, :]
    matrix = matrix * ix[:, np.newaxis]
    identity = identity * ix
  if relative_matrix_epsilon:
    _, max_ev, _ = power_iteration(matrix, 100, error_tolerance, padding_start)
  else:
    max_ev = F32(1.0)
  ridge = F32(F32(ridge_epsilon) * np.maximum(F32(max_ev), F32(error_tolerance)))
  reg = matrix + ridge * identity
  if np.all(np.isfinite(reg)):
    e, u = eigh_fn(reg)  # DS:1071
  else:
    e, u = np.full(d, np.nan, F32), np.full((d, d), np.nan, F32)
  e, u = e.astype(F32), u.astype(F32)
  if ix is not None:
    e = e * ix[::-1]
  recovered = np.matmul(np.array(u.T), np.matmul(reg, u))
  eig_error = recovered - np.diag(e)
  if ix is not None:
    eig_error = eig_error * ix[::-1]
  error = F32(np.max(np.abs(eig_error)))
  with np.errstate(all="ignore"):
    inv_e = np.where(e == 0.0, F32(0.0), np.power(np.maximum(e, ridge), alpha)).astype(F32)
  ps = d if padding_start is None else padding_start
  if compression_rank < 0:
    inv_e = np.roll(inv_e, -(d - ps))
    u = np.roll(u, -(d - ps), axis=1)
  else:
    inv_e = inv_e[::-1]
    u = u[:, ::-1]
  r = abs(compression_rank)
  keep_e, to_avg = inv_e[:r], inv_e[r:]
  num = ps - r
  const = F32(np.sum(to_avg) / (num if num > 0 else 1.0))
  val = fd_low_rank_pack(u[:, :r], np.zeros(r, F32), keep_e, const, 0.0, False, r)
  if padding_start is not None and padding_start == 0:
    val = np.zeros_like(val)
    error = F32(0.0)
  return val, float(error)


def frequent_directions_update(g, axis, lapack="f32"):
  """DS:1497-1505: zero-padded R^T from qr(x^T, mode='r') (LAPACK sgeqrf); R R^T = x x^T."""
  g = np.asarray(g, F32)
  x = np.reshape(np.moveaxis(g, axis, 0), (g.shape[axis], -1))
  r = _lapack_fns(lapack)[2](np.ascontiguousarray(x.T)).T
  return np.pad(r, ((0, 0), (0, x.shape[0] - r.shape[1]))).astype(F32)


def fd_update_root(new_grad, p, rank, ridge_epsilon=1e-6, error_tolerance=1e-6,
                   relative_matrix_epsilon=True, decay=1.0, padding_start=None, prev=None,
                   lapack="f32"):
  """DS:1123-1290 (without FDDiagnostics), SVD by LAPACK sgesdd (float32) as in the reference."""
  new_grad = np.array(new_grad, F32)
  max_size = new_grad.shape[0]
  ps = max_size if padding_start is None else padding_start
  sketch, fwd, _, _, tail, _ = fd_low_rank_unpack(prev, rank)
  max_ev = fwd[0] if relative_matrix_epsilon else F32(1.0)
  ridge = F32(F32(ridge_epsilon) * np.maximum(F32(max_ev), F32(error_tolerance)))
  act_d = ps > np.arange(max_size)
  act_r = ps > np.arange(rank)
  sketch = sketch * act_d[:, None] * act_r
  fwd = (fwd + ridge) * act_r
  weighted = (sketch * np.sqrt(fwd)).astype(F32)
  padded = new_grad * act_d * act_d[:, None]
  updated = np.concatenate([F32(np.sqrt(decay)) * weighted, padded], axis=1).astype(F32)
  u, s, _ = _lapack_fns(lapack)[1](updated)  # DS:1193
  cutoff = s[rank]
  rho = cutoff ** 2
  top = s[:rank]
  deflated = (top - cutoff) * (top + cutoff)
  eigvecs = u[:, :rank].copy()
  tail = F32(tail * F32(decay))
  new_tail = F32(tail + rho)
  alpha = F32(-1.0 / int(p))
  with np.errstate(all="ignore"):
    # np.power (the ufunc loop jnp.where(..., new_tail**alpha) runs over the stand-in), not the
    # scalar `**`, which NumPy evaluates with a different powf and can differ by one ulp
    new_const = F32(0.0) if new_tail <= 0 else F32(np.power(np.asarray(new_tail, F32), alpha))
  new_tail = F32(0.0) if new_tail <= 0 else new_tail
  deflated = np.where(deflated <= 0, F32(0.0), deflated).astype(F32)
  eigvecs = eigvecs * (deflated > 0)
  norms = np.linalg.norm(eigvecs, axis=0)
  safe = (0.99 <= norms) & (norms <= 1.01)
  eigvecs = eigvecs * safe
  deflated = deflated * safe
  eigvecs = eigvecs / np.where(safe, norms, 1.0)
  pad_ix = np.arange(max_size) >= ps
  mass = np.linalg.norm(eigvecs * pad_ix[:, None], axis=0, ord=1)
  has_pad = mass > 0.01
  eigvecs = eigvecs * (1 - has_pad)
  deflated = deflated * (1 - has_pad)
  up = (np.square(top) + tail) * (deflated > 0.0)
  up = np.where(up <= 0, F32(0.0), up)
  with np.errstate(all="ignore"):
    inverted = np.where(up <= 0, F32(0.0), up ** alpha).astype(F32)
  has_zeros = bool(np.any(deflated <= 0)) or bool(new_tail <= 0)
  val = fd_low_rank_pack(eigvecs.astype(F32), deflated.astype(F32), inverted, new_const,
                         new_tail, has_zeros, rank)
  if padding_start is not None and padding_start == 0:
    val = np.zeros_like(val)
  return val


def precondition_block_low_rank(g, pc, rank):
  """DS:1690-1705 for one axis: low-rank + constant application, result rolled."""
  eigvecs, _, eigvals, const, _, skip = fd_low_rank_unpack(pc, rank)
  nd = g.ndim
  basis = np.tensordot(g, eigvecs, axes=[[0], [0]])
  comp = np.tensordot(basis, eigvecs, axes=[[nd - 1], [1]])
  gr = np.transpose(g, tuple(range(1, nd)) + (0,))
  complement = gr - comp
  scaled = np.tensordot(basis * eigvals, eigvecs, axes=[[nd - 1], [1]])
  new_g = const * complement + scaled
  return (gr if skip else new_g).astype(F32)
