"""True float32 LAPACK for the sites the reference delegates to jax / XLA.  TEST INFRASTRUCTURE ONLY.

With ``jax_enable_x64`` off (DS:35-38: ``_MAT_INV_PTH_ROOT_DTYPE = jnp.float64`` degrades to float32)
the reference's JAX CPU path lowers

    jnp.linalg.eigh      DS:1007, DS:1071   -> LAPACK ``ssyevd`` (jobz = 'V', uplo = 'L', on (x + x^T) / 2)
    jnp.linalg.eigvalsh  DS:304             -> the same ``ssyevd`` call, vectors discarded
    jnp.linalg.svd       DS:1193            -> LAPACK ``sgesdd`` (jobz = 'S' for full_matrices=False)
    jnp.linalg.qr        DS:1502 (mode 'r') -> LAPACK ``sgeqrf``, upper triangle of the result

``numpy.linalg.eigh / svd / qr`` are NOT those routines for float32 inputs: NumPy converts to float64,
runs ``dsyevd`` / ``dgesdd`` / ``dgeqrf`` and rounds the result (``_commonType`` always computes in
double), so its "float32" answers are 2-4 decades more accurate than what the reference computes
(512 x 512, cond 1e4, p = 2: root error vs float64 5.6e-4 with ``ssyevd``, 3.8e-8 with NumPy).  Until
round 6 the oracle and the golden generator used NumPy here; both now call the single-precision
routines through ``scipy.linalg.lapack`` (SciPy's own OpenBLAS / LAPACK build, same image here and
on the GPU box).  The ``*64`` functions are the float64-internal versions, kept as the ACCURACY
YARDSTICK beside the parity target (fixtures carry both).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import lapack as _lp

F32 = np.float32


def _f32(a):
  a = np.asarray(a)
  assert a.dtype == F32, a.dtype
  return a


def eigh32(a):
  """(w ascending, v) of a symmetric float32 matrix by ``ssyevd`` — what jnp.linalg.eigh runs on CPU
  (jax symmetrises its input first: ``symmetrize_input=True``)."""
  a = _f32(a)
  s = ((a + a.T) * F32(0.5)).astype(F32)
  w, v, info = _lp.ssyevd(s, compute_v=1, lower=1)
  if info != 0:  # XLA reports failure as NaNs, never raises inside a step
    return np.full(a.shape[0], np.nan, F32), np.full(a.shape, np.nan, F32)
  assert w.dtype == F32 and v.dtype == F32
  return w, np.ascontiguousarray(v)


def eigvalsh32(a):
  return eigh32(a)[0]


def svd32(a):
  """(u, s, vt) with full_matrices=False by ``sgesdd`` (jobz 'S')."""
  a = _f32(a)
  u, s, vt, info = _lp.sgesdd(a, compute_uv=1, full_matrices=0)
  if info != 0:
    k = min(a.shape)
    return (np.full((a.shape[0], k), np.nan, F32), np.full(k, np.nan, F32),
            np.full((k, a.shape[1]), np.nan, F32))
  assert u.dtype == F32 and s.dtype == F32 and vt.dtype == F32
  return np.ascontiguousarray(u), s, np.ascontiguousarray(vt)


def qr_r32(a):
  """R of ``qr(a, mode='r')`` by ``sgeqrf``: [min(m, n), n] upper triangular, LAPACK's signs."""
  a = _f32(a)
  qr, _tau, _work, info = _lp.sgeqrf(a)
  assert info == 0 and qr.dtype == F32
  k = min(a.shape)
  return np.triu(qr[:k]).astype(F32)


# float64-internal versions (what NumPy computes for float32 inputs): accuracy yardstick only
def eigh64(a):
  w, v = np.linalg.eigh(_f32(a))
  return w.astype(F32), v.astype(F32)


def svd64(a):
  u, s, vt = np.linalg.svd(_f32(a), full_matrices=False)
  return u.astype(F32), s.astype(F32), vt.astype(F32)


def qr_r64(a):
  return np.linalg.qr(_f32(a), mode="r").astype(F32)
