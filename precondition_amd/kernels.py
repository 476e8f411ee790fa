"""Torch-tensor front end of the C-ABI (include/ps_api.h).

Torch is plumbing here: it owns device memory and the current HIP stream; all
arithmetic happens in libprecondition_amd.so.  Function names and argument
meaning follow the reference's public helpers
(precondition/distributed_shampoo.py: power_iteration DS:595, mat_power DS:655,
matrix_inverse_pth_root DS:702, gram_weighted_update DS:1440).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import GemmDesc, StatsDesc, TransformConfig, TransformDesc, check, lib


def _require_gpu(t: torch.Tensor, what: str):
  if not t.is_cuda:
    raise _lib.PsError(
        f"{what}: expected a tensor on an MI355X device, got {t.device}; "
        "precondition_amd has no CPU path.")
  if t.dtype != torch.float32:
    raise TypeError(f"{what}: expected float32, got {t.dtype}")


def _stream() -> int:
  """HIP stream of the CURRENT device; every entry point below runs under
  `_device_guarded`, which makes the tensors' device current first."""
  return torch.cuda.current_stream().cuda_stream


def _first_cuda_device(obj, depth=0):
  if isinstance(obj, torch.Tensor):
    return obj.device if obj.is_cuda else None
  if depth < 3 and isinstance(obj, (list, tuple)):
    for x in obj:
      d = _first_cuda_device(x, depth + 1)
      if d is not None:
        return d
  if depth < 3 and isinstance(obj, dict):
    for x in obj.values():
      d = _first_cuda_device(x, depth + 1)
      if d is not None:
        return d
  return None


def _device_guarded(fn):
  """Runs `fn` with the device of its first device-tensor argument current, so that the
  stream handed to the library (_stream()), the workspaces and the outputs all belong to
  the device the pointers live on (kernels enqueued on a cuda:0 stream with cuda:1
  pointers fault or race with the caching allocator).  The library itself serves one
  device per process (PS_EDEVICE otherwise)."""
  import functools

  @functools.wraps(fn)
  def wrapper(*args, **kwargs):
    dev = _first_cuda_device(args) or _first_cuda_device(list(kwargs.values()))
    if dev is None:
      return fn(*args, **kwargs)
    with torch.cuda.device(dev):
      return fn(*args, **kwargs)

  return wrapper


def _same_device(tensors, what: str):
  devs = {t.device for t in tensors if isinstance(t, torch.Tensor)}
  if len(devs) > 1:
    raise ValueError(f"{what}: all tensors of one call must live on one device, got "
                     f"{sorted(str(d) for d in devs)}")


def _i32(xs) -> np.ndarray:
  return np.ascontiguousarray(np.asarray(xs, dtype=np.int32))


def _ptrs(ts: Sequence[torch.Tensor]) -> np.ndarray:
  return np.ascontiguousarray(
      np.asarray([t.data_ptr() for t in ts], dtype=np.uint64))


def _workspace(nbytes: int, device) -> torch.Tensor:
  """Scratch for one library call.  Allocated on the current stream, which is also the
  stream the call is enqueued on (_stream()), so the caching allocator's stream-ordered
  reuse is already safe once the tensor is dropped: no record_stream needed."""
  return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _as_2d_ld(t: torch.Tensor) -> int:
  """Leading dimension of a row-major 2-D view (last stride must be 1)."""
  assert t.dim() == 2
  if t.shape[1] > 1 and t.stride(1) != 1:
    raise ValueError("matrix rows must be contiguous")
  return int(t.stride(0)) if t.shape[0] > 1 else max(int(t.shape[1]), 1)


# ---------------------------------------------------------------------------
# inverse p-th root
# ---------------------------------------------------------------------------
@_device_guarded
def matrix_inverse_pth_root_batched(
    matrices: Sequence[torch.Tensor],
    ps: Sequence[int],
    padding_starts: Optional[Sequence[int]] = None,
    num_iters: int = 100,
    ridge_epsilon: float = 1e-6,
    error_tolerance: float = 1e-6,
    relative_matrix_epsilon: bool = True,
    eigh: bool = False,
    out: Optional[Sequence[torch.Tensor]] = None,
    max_ev: Optional[torch.Tensor] = None,
    symmetry="verify",
    options=None,
) -> Tuple[List[torch.Tensor], torch.Tensor]:
  """vmap(matrix_inverse_pth_root) over independent blocks (DS:2742-2744).

  Returns (roots, metrics[batch, 8]); metrics columns are the PS_M_* indices
  (0..4 = TrainingMetrics fields of DS:902-907).  Blocks may differ in size.
  `max_ev` (float32 device tensor [batch]): the largest eigenvalue is given instead of
  being found by the power iteration (the lobpcg branch, DS:813-817).
  `symmetry`: 'verify' | 'assume' | 'general' (_lib.symmetry_code): the reference takes
  any square matrix; exactly symmetric blocks get the half-work symmetric products.
  `options`: per-call modes (dict, see _lib.make_options / ps_options in include/ps_api.h):
  product arithmetic (the reference's `precision`), accumulation, averaged steps, the
  iteration-count hint of the previous recompute, execution, power-iteration execution.
  """
  stacked = isinstance(matrices, torch.Tensor)
  if stacked:
    # The reference's own batched form, xs[b, n, n] (DS:2742-2744): one stacked tensor in, one
    # stacked tensor out -- the pointer / size tables are arithmetic on the base address instead of
    # a Python loop over b tensors (0.6 ms of host time per call at b = 256, which a recompute
    # issued back to back with the previous one cannot hide).
    xs = matrices
    _require_gpu(xs, "matrix_inverse_pth_root")
    if xs.dim() != 3 or xs.shape[1] != xs.shape[2] or xs.stride(2) != 1 or xs.dtype != torch.float32:
      raise ValueError(f"expected a float32 [b, n, n] tensor with contiguous rows, got {tuple(xs.shape)}")
    batch, nn = int(xs.shape[0]), int(xs.shape[1])
    if batch == 0:
      return xs.new_empty((0, nn, nn)), torch.empty((0, _lib.PS_METRICS_STRIDE), dtype=torch.float32)
    dev = xs.device
    if out is None:
      out = torch.empty((batch, nn, nn), dtype=torch.float32, device=dev)
    if (not isinstance(out, torch.Tensor) or tuple(out.shape) != (batch, nn, nn) or out.stride(2) != 1
        or out.dtype != torch.float32 or out.device != dev):
      raise ValueError("out must be a float32 [b, n, n] tensor on the same device")
    n = np.full(batch, nn, np.int32)
    lda = np.full(batch, int(xs.stride(1)), np.int32)
    ldo = np.full(batch, int(out.stride(1)), np.int32)
    steps = np.arange(batch, dtype=np.uint64)
    a_ptrs = np.uint64(xs.data_ptr()) + steps * np.uint64(4 * int(xs.stride(0)))
    o_ptrs = np.uint64(out.data_ptr()) + steps * np.uint64(4 * int(out.stride(0)))
    p = _i32(list(ps))
    pad = None if padding_starts is None else _i32(list(padding_starts))
    if len(p) != batch or (pad is not None and len(pad) != batch):
      raise ValueError("ps / padding_starts must hold one value per block")
  else:
    batch = len(matrices)
    if batch == 0:
      return [], torch.empty((0, _lib.PS_METRICS_STRIDE), dtype=torch.float32)
    dev = matrices[0].device
    for m in matrices:
      _require_gpu(m, "matrix_inverse_pth_root")
      if m.dim() != 2 or m.shape[0] != m.shape[1]:
        raise ValueError(f"expected square matrices, got {tuple(m.shape)}")
    _same_device(list(matrices) + (list(out) if out is not None else []),
                 "matrix_inverse_pth_root")
    n = _i32([m.shape[0] for m in matrices])
    lda = _i32([_as_2d_ld(m) for m in matrices])
    p = _i32(list(ps))
    pad = None if padding_starts is None else _i32(list(padding_starts))
    if out is None:
      out = [torch.empty((int(k), int(k)), dtype=torch.float32, device=dev)
             for k in n]
    ldo = _i32([_as_2d_ld(o) for o in out])
    a_ptrs, o_ptrs = _ptrs(matrices), _ptrs(out)
  metrics = torch.empty((batch, _lib.PS_METRICS_STRIDE), dtype=torch.float32,
                        device=dev)
  L = lib()
  pad_ptr = None if pad is None else pad.ctypes.data
  popt, _keep = _lib.make_options(options)
  if popt.iters_hint and len(_keep[0]) != batch:
    raise ValueError(f"iters_hint must hold one value per block ({batch}), got {len(_keep[0])}")
  if eigh:
    nbytes = L.ps_eigh_root_workspace_bytes(batch, n.ctypes.data)
    ws = _workspace(nbytes, dev)
    rc = L.ps_eigh_root_batched_opt_f32(
        _stream(), a_ptrs.ctypes.data, n.ctypes.data, lda.ctypes.data,
        p.ctypes.data, pad_ptr, batch, ridge_epsilon, error_tolerance,
        int(relative_matrix_epsilon), o_ptrs.ctypes.data, ldo.ctypes.data,
        metrics.data_ptr(), ws.data_ptr(), ws.numel(), C.byref(popt))
    check(rc, "ps_eigh_root_batched_opt_f32")
  else:
    nbytes = L.ps_newton_root_workspace_bytes(batch, n.ctypes.data,
                                              p.ctypes.data, pad_ptr)
    if nbytes == 0:
      raise _lib.PsError("ps_newton_root_workspace_bytes: unsupported exponent")
    ws = _workspace(nbytes, dev)
    iters = C.c_int32(0)
    mev_ptr = None
    if max_ev is not None:
      if not relative_matrix_epsilon:
        raise ValueError("max_ev only applies to the relative-epsilon form")
      mev = max_ev.to(torch.float32).contiguous()
      if not mev.is_cuda or mev.numel() != batch:
        raise ValueError("max_ev must be a device tensor with one value per block")
      mev_ptr = mev.data_ptr()
    rc = L.ps_newton_root_batched_opt_f32(
        _stream(), a_ptrs.ctypes.data, n.ctypes.data, lda.ctypes.data,
        p.ctypes.data, pad_ptr, batch, num_iters, ridge_epsilon,
        error_tolerance, int(relative_matrix_epsilon), mev_ptr, _lib.symmetry_code(symmetry),
        o_ptrs.ctypes.data, ldo.ctypes.data, metrics.data_ptr(), ws.data_ptr(), ws.numel(),
        C.addressof(iters), C.byref(popt))
    check(rc, "ps_newton_root_batched_opt_f32")
  return (out if stacked else list(out)), metrics


def products_for_precision(precision) -> str:
  """The reference's `precision` kwarg (jax.lax.Precision, DS:599, 708, 1883) -> ps_options.products:
  HIGHEST (the reference's default) / None -> 'f32' (exact float32 MFMA, the parity path);
  HIGH -> 'bf16x6' (three-way bf16 split, what XLA calls bf16_3x... float32-faithful);
  DEFAULT -> 'bf16x3'.  Accepts the enum member (anything with a .name), its name in any case,
  or one of the ps_options spellings."""
  if precision is None:
    return "f32"
  name = getattr(precision, "name", precision)
  if not isinstance(name, str):
    raise ValueError(f"precision must be None, a jax.lax.Precision member or a string, got {precision!r}")
  key = name.strip().lower()
  table = {"highest": "f32", "float32": "f32", "fp32": "f32", "f32": "f32",
           "high": "bf16x6", "bfloat16_3x": "bf16x6", "bf16_3x": "bf16x6", "bf16x6": "bf16x6",
           "default": "bf16x3", "bfloat16": "bf16x3", "bf16": "bf16x3", "bf16x3": "bf16x3"}
  if key not in table:
    raise ValueError(f"unknown precision {precision!r}")
  return table[key]


def matrix_inverse_pth_root(matrix: torch.Tensor, p: int, num_iters: int = 100,
                            ridge_epsilon: float = 1e-6,
                            error_tolerance: float = 1e-6, precision=None,
                            relative_matrix_epsilon: bool = True,
                            lobpcg_topk_precondition: int = 0,
                            lobpcg_max_iter: int = 0,
                            padding_start: Optional[int] = None, prev=None,
                            eigh: bool = False):
  """Single-matrix form with the reference's signature (DS:702-715).  Returns
  (root, TrainingMetrics)."""
  from .state import TrainingMetrics
  del prev
  options = {"products": products_for_precision(precision)}
  if lobpcg_topk_precondition:
    if eigh:
      raise ValueError("lobpcg_topk_precondition applies to the Newton branch")
    from . import deflation
    roots, m, diags = deflation.matrix_inverse_pth_root_deflated_batched(
        [matrix], [p], None if padding_start is None else [padding_start],
        topk=lobpcg_topk_precondition, max_iter=lobpcg_max_iter, num_iters=num_iters,
        ridge_epsilon=ridge_epsilon, error_tolerance=error_tolerance,
        relative_matrix_epsilon=relative_matrix_epsilon)
    return roots[0], TrainingMetrics(
        inverse_pth_root_errors=m[0, 0], inverse_pth_root_iters=m[0, 1],
        final_error_ratio=m[0, 2], max_eigen_value=m[0, 3], total_retries=m[0, 4],
        **diags[0])
  roots, m = matrix_inverse_pth_root_batched(
      [matrix], [p], None if padding_start is None else [padding_start],
      num_iters=num_iters, ridge_epsilon=ridge_epsilon,
      error_tolerance=error_tolerance,
      relative_matrix_epsilon=relative_matrix_epsilon, eigh=eigh, options=options)
  return roots[0], TrainingMetrics(
      inverse_pth_root_errors=m[0, 0], inverse_pth_root_iters=m[0, 1],
      final_error_ratio=m[0, 2], max_eigen_value=m[0, 3], total_retries=m[0, 4])


def matrix_inverse_pth_root_deflated_batched(*args, **kwargs):
  """The lobpcg_topk_precondition branch (DS:787-812, 889-928), see deflation.py."""
  from . import deflation
  return deflation.matrix_inverse_pth_root_deflated_batched(*args, **kwargs)


@_device_guarded
def eigh_batched(matrices: Sequence[torch.Tensor], options=None):
  """jnp.linalg.eigh for a batch of symmetric matrices.  Returns (eigenvalues ascending [n],
  eigenvectors [n, n] in columns) per matrix, LAPACK order; signs of eigenvectors are arbitrary.
  `options`: eigh_solver / eigh_sweep_tol / eigh_streams of _lib.make_options (default 'auto':
  the tridiagonalisation path where the matrix is positive definite and well conditioned, the
  Jacobi solvers -- accurate relative to every eigenvalue -- otherwise)."""
  batch = len(matrices)
  dev = matrices[0].device
  for m in matrices:
    _require_gpu(m, "eigh")
  mats = [m.contiguous() for m in matrices]
  n = _i32([m.shape[0] for m in mats])
  lda = _i32([_as_2d_ld(m) for m in mats])
  evals = [torch.empty((int(k),), dtype=torch.float32, device=dev) for k in n]
  evecs = [torch.empty((int(k), int(k)), dtype=torch.float32, device=dev) for k in n]
  ldv = _i32([max(int(k), 1) for k in n])
  L = lib()
  ws = _workspace(L.ps_eigh_root_workspace_bytes(batch, n.ctypes.data), dev)
  a_ptrs, e_ptrs, v_ptrs = _ptrs(mats), _ptrs(evals), _ptrs(evecs)
  popt, _keep = _lib.make_options(options)
  rc = L.ps_eigh_batched_opt_f32(_stream(), a_ptrs.ctypes.data, n.ctypes.data,
                                 lda.ctypes.data, batch, e_ptrs.ctypes.data,
                                 v_ptrs.ctypes.data, ldv.ctypes.data, ws.data_ptr(),
                                 ws.numel(), C.byref(popt))
  check(rc, "ps_eigh_batched_opt_f32")
  sorted_n = L.ps_eigh_sorted_max_n()  # small matrices come back ascending already
  out_e, out_v = [], []
  for e, v in zip(evals, evecs):
    if e.shape[0] <= sorted_n:
      out_e.append(e)
      out_v.append(v)
      continue
    order = torch.argsort(e)  # pure permutation (data movement, no arithmetic)
    out_e.append(e[order])
    out_v.append(v[:, order])
  return out_e, out_v


@_device_guarded
def power_iteration(matrix: torch.Tensor, num_iters: int = 100,
                    error_tolerance: float = 1e-6,
                    padding_start: Optional[int] = None, symmetry="verify"):
  """DS:595-652.  Returns (eigenvector, eigenvalue) as device tensors.  Any square
  matrix (the reference's is a plain mat-vec loop); see `symmetry` above."""
  _require_gpu(matrix, "power_iteration")
  n = int(matrix.shape[-1])
  dev = matrix.device
  na, lda = _i32([n]), _i32([_as_2d_ld(matrix)])
  pad = None if padding_start is None else _i32([padding_start])
  lam = torch.empty(1, dtype=torch.float32, device=dev)
  its = torch.empty(1, dtype=torch.int32, device=dev)
  v = torch.zeros((1, n), dtype=torch.float32, device=dev)
  L = lib()
  ws = _workspace(L.ps_power_iteration_workspace_bytes(1, na.ctypes.data), dev)
  a_ptrs = _ptrs([matrix])
  rc = L.ps_power_iteration_batched_f32(
      _stream(), a_ptrs.ctypes.data, na.ctypes.data, lda.ctypes.data,
      None if pad is None else pad.ctypes.data, 1, num_iters, error_tolerance,
      lam.data_ptr(), its.data_ptr(), v.data_ptr(), n, _lib.symmetry_code(symmetry),
      ws.data_ptr(), ws.numel())
  check(rc, "ps_power_iteration_batched_f32")
  return v[0], lam[0]


@_device_guarded
def power_iteration_batched(matrices: Sequence[torch.Tensor], num_iters=100,
                            error_tolerance=1e-6, padding_starts=None, symmetry="verify",
                            options=None):
  """Returns (lambda[batch], iters[batch]).  `options`: power_iteration / pi_timeout_ms of
  _lib.make_options."""
  batch = len(matrices)
  dev = matrices[0].device
  for m in matrices:
    _require_gpu(m, "power_iteration")
  n = _i32([m.shape[0] for m in matrices])
  lda = _i32([_as_2d_ld(m) for m in matrices])
  pad = None if padding_starts is None else _i32(list(padding_starts))
  lam = torch.empty(batch, dtype=torch.float32, device=dev)
  its = torch.empty(batch, dtype=torch.int32, device=dev)
  L = lib()
  ws = _workspace(L.ps_power_iteration_workspace_bytes(batch, n.ctypes.data), dev)
  a_ptrs = _ptrs(matrices)
  popt, _keep = _lib.make_options(options)
  rc = L.ps_power_iteration_batched_opt_f32(
      _stream(), a_ptrs.ctypes.data, n.ctypes.data, lda.ctypes.data,
      None if pad is None else pad.ctypes.data, batch, num_iters,
      error_tolerance, lam.data_ptr(), its.data_ptr(), None, 0,
      _lib.symmetry_code(symmetry), ws.data_ptr(), ws.numel(), C.byref(popt))
  check(rc, "ps_power_iteration_batched_opt_f32")
  return lam, its


@_device_guarded
def mat_power(mat_m: torch.Tensor, p: int) -> torch.Tensor:
  """DS:655-678: M^p with the reference's multiplication order."""
  _require_gpu(mat_m, "mat_power")
  n = int(mat_m.shape[0])
  out = torch.empty((n, n), dtype=torch.float32, device=mat_m.device)
  L = lib()
  ws = _workspace(L.ps_mat_power_workspace_bytes(n, int(p)), mat_m.device)
  rc = L.ps_mat_power_f32(_stream(), mat_m.data_ptr(), n, _as_2d_ld(mat_m),
                          int(p), out.data_ptr(), n, ws.data_ptr(), ws.numel())
  check(rc, "ps_mat_power_f32")
  return out


@_device_guarded
def matmul(a: torch.Tensor, b: torch.Tensor, transa: bool = False,
           transb: bool = False) -> torch.Tensor:
  """op(a) @ op(b) in float32 on the MFMA core (2-D, or 3-D batched)."""
  _require_gpu(a, "matmul")
  _require_gpu(b, "matmul")
  squeeze = a.dim() == 2
  a3 = (a.unsqueeze(0) if squeeze else a).contiguous()
  b3 = (b.unsqueeze(0) if squeeze else b).contiguous()
  bt = a3.shape[0]
  m, k = (a3.shape[2], a3.shape[1]) if transa else (a3.shape[1], a3.shape[2])
  n = b3.shape[1] if transb else b3.shape[2]
  kb = b3.shape[2] if transb else b3.shape[1]
  if kb != k or b3.shape[0] != bt:
    raise ValueError(f"matmul shape mismatch: {tuple(a.shape)} x {tuple(b.shape)}")
  c = torch.empty((bt, m, n), dtype=torch.float32, device=a.device)
  rc = lib().ps_gemm_f32(_stream(), int(transa), int(transb), a3.data_ptr(),
                         b3.data_ptr(), c.data_ptr(), m, n, k, a3.shape[2],
                         b3.shape[2], n, bt, a3.shape[1] * a3.shape[2],
                         b3.shape[1] * b3.shape[2], m * n)
  check(rc, "ps_gemm_f32")
  return c[0] if squeeze else c


_GDESC_DT = np.dtype([("a", "u8"), ("b", "u8"), ("c", "u8"), ("m", "i4"), ("n", "i4"),
                      ("k", "i4"), ("transa", "i4"), ("transb", "i4"), ("lda", "i8"),
                      ("ldb", "i8"), ("ldc", "i8")], align=True)


def _ld_2d(shapes: np.ndarray, strides: np.ndarray) -> np.ndarray:
  """_as_2d_ld for [n, 2] shape / stride arrays.  (Callers convert torch.Size with tuple():
  np.array over a list of torch.Size objects is ~25x slower than over tuples.)"""
  if np.any((shapes[:, 1] > 1) & (strides[:, 1] != 1)):
    raise ValueError("matrix rows must be contiguous")
  return np.where(shapes[:, 0] > 1, strides[:, 0], np.maximum(shapes[:, 1], 1))


class GemmPlan:
  """A grouped product whose task tables are built and uploaded ONCE (ps_gemm_grouped_plan_*):
  `launch()` is one or two kernel launches with no host-side table building.  The operand tensors
  of `items` (same form as gemm_grouped) are kept alive by the plan; their contents may change
  between launches, their storage may not."""

  def __init__(self, items):
    self._keep = [t for it in items for t in it[:3]]
    tbl, dev = _gemm_desc_table(items)
    self._dev = dev
    descs = C.cast(tbl.ctypes.data, C.POINTER(GemmDesc))
    L = lib()
    self._ws = _workspace(L.ps_gemm_grouped_workspace_bytes(descs, len(items)), dev)
    h = C.c_void_p()
    with torch.cuda.device(dev):
      rc = L.ps_gemm_grouped_plan_create(_stream(), descs, len(items), self._ws.data_ptr(),
                                         self._ws.numel(), C.byref(h))
    check(rc, "ps_gemm_grouped_plan_create")
    self._h = h

  def launch(self):
    with torch.cuda.device(self._dev):
      check(lib().ps_gemm_grouped_plan_launch(_stream(), self._h), "ps_gemm_grouped_plan_launch")

  def __del__(self):
    h, self._h = getattr(self, "_h", None), None
    if h:
      try:
        lib().ps_gemm_grouped_plan_destroy(h)
      except Exception:  # pylint: disable=broad-except
        pass


@_device_guarded
def gemm_grouped(items):
  """items: list of (a, b, c, transa, transb) with 2-D row-contiguous (possibly
  strided) tensors; c = op(a) @ op(b) for all of them in one launch per layout pair.
  The descriptor table is filled column-wise with NumPy (a parameter tree has hundreds
  of blocks: per-item ctypes stores were the cost of a step)."""
  if not items:
    return
  tbl, dev = _gemm_desc_table(items)
  n_items = len(items)
  descs = C.cast(tbl.ctypes.data, C.POINTER(GemmDesc))
  L = lib()
  ws = _workspace(L.ps_gemm_grouped_workspace_bytes(descs, n_items), dev)
  rc = L.ps_gemm_grouped_f32(_stream(), descs, n_items, ws.data_ptr(), ws.numel())
  check(rc, "ps_gemm_grouped_f32")


def _gemm_desc_table(items):
  """(descriptor table, device) of a gemm_grouped item list."""
  n_items = len(items)
  dev = items[0][0].device
  A = [it[0] for it in items]
  B = [it[1] for it in items]
  Cm = [it[2] for it in items]
  for t in A + B + Cm:
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2:
      _require_gpu(t, "gemm_grouped")
      raise ValueError("gemm_grouped expects 2-D tensors")
  _same_device(A + B + Cm, "gemm_grouped")
  sa = np.array([tuple(t.shape) for t in A], np.int64)
  sb = np.array([tuple(t.shape) for t in B], np.int64)
  sc = np.array([tuple(t.shape) for t in Cm], np.int64)
  ta = np.array([bool(it[3]) for it in items])
  tb = np.array([bool(it[4]) for it in items])
  m = np.where(ta, sa[:, 1], sa[:, 0])
  k = np.where(ta, sa[:, 0], sa[:, 1])
  n = np.where(tb, sb[:, 0], sb[:, 1])
  kb = np.where(tb, sb[:, 1], sb[:, 0])
  bad = (kb != k) | (sc[:, 0] != m) | (sc[:, 1] != n)
  if bad.any():
    raise ValueError(f"gemm_grouped shape mismatch in item {int(np.argmax(bad))}")
  tbl = np.zeros(n_items, _GDESC_DT)
  assert _GDESC_DT.itemsize == C.sizeof(GemmDesc)
  tbl["a"] = [t.data_ptr() for t in A]
  tbl["b"] = [t.data_ptr() for t in B]
  tbl["c"] = [t.data_ptr() for t in Cm]
  tbl["m"], tbl["n"], tbl["k"], tbl["transa"], tbl["transb"] = m, n, k, ta, tb
  tbl["lda"] = _ld_2d(sa, np.array([t.stride() for t in A], np.int64))
  tbl["ldb"] = _ld_2d(sb, np.array([t.stride() for t in B], np.int64))
  tbl["ldc"] = _ld_2d(sc, np.array([t.stride() for t in Cm], np.int64))
  return tbl, dev


class TiledBf16:
  """Left operand of the bf16 products in a blocked layout (to_bf16(..., tiled=True | "frag")):
  `hi` / `lo` are flat bfloat16 buffers.  layout 1: 128 x 32 tiles contiguous (gemm_bf16_grouped,
  ceil(rows / 128) * 128 * cols elements); layout 2: fragment-major (fd_filter_round / fd_cy_step:
  the kilobyte one MFMA consumes is contiguous; rows * cols elements, both multiples of 64)."""

  def __init__(self, hi, lo, rows, cols, layout=1, lo2=None):
    self.hi, self.lo, self.rows, self.cols, self.layout = hi, lo, rows, cols, layout
    self.lo2 = lo2      # third plane (x = hi + lo + lo2 to float32 accuracy): to_bf16(..., tiled="frag3")


@_device_guarded
def to_bf16(x: torch.Tensor, split: bool = False, transpose: bool = False, tiled: bool = False):
  """float32 [r, c] -> bfloat16 operand(s) for gemm_bf16_grouped: (hi, lo | None), each
  [r, c] (or [c, r] with transpose), round to nearest even; lo = bf16(x - hi).  tiled: a
  TiledBf16 (128 x 32 tiles contiguous: the layout a big left operand should have, it is streamed
  from HBM once per product; needs c % 32 == 0)."""
  _require_gpu(x, "to_bf16")
  if x.dim() != 2:
    raise ValueError("to_bf16 expects a 2-D tensor")
  r, c = int(x.shape[0]), int(x.shape[1])
  if tiled == "frag3":   # fragment-major hi / lo / lo2 (the six-product form of fd_cx6)
    if transpose or not split or c % 64 != 0 or r % 64 != 0:
      raise ValueError("fragment-major bf16 operands: split, no transpose, rows and columns multiples of 64")
    p = [torch.empty((r * c,), dtype=torch.bfloat16, device=x.device) for _ in range(3)]
    rc = lib().ps_convert_f32_to_bf16x3_frag(_stream(), x.data_ptr(), p[0].data_ptr(), p[1].data_ptr(),
                                             p[2].data_ptr(), r, c, _as_2d_ld(x))
    check(rc, "ps_convert_f32_to_bf16x3_frag")
    return TiledBf16(p[0], p[1], r, c, layout=2, lo2=p[2])
  if tiled == "frag":
    if transpose or c % 64 != 0 or r % 64 != 0:
      raise ValueError("fragment-major bf16 operands: no transpose, rows and columns multiples of 64")
    hi = torch.empty((r * c,), dtype=torch.bfloat16, device=x.device)
    lo = torch.empty((r * c,), dtype=torch.bfloat16, device=x.device) if split else None
    rc = lib().ps_convert_f32_to_bf16(_stream(), x.data_ptr(), hi.data_ptr(),
                                      lo.data_ptr() if split else None, r, c, _as_2d_ld(x), c, 3)
    check(rc, "ps_convert_f32_to_bf16")
    return TiledBf16(hi, lo, r, c, layout=2)
  if tiled:
    if transpose or c % 32 != 0:
      raise ValueError("tiled bf16 operands: no transpose, columns a multiple of 32")
    numel = ((r + 127) // 128) * 128 * c
    hi = torch.empty((numel,), dtype=torch.bfloat16, device=x.device)
    lo = torch.empty((numel,), dtype=torch.bfloat16, device=x.device) if split else None
    rc = lib().ps_convert_f32_to_bf16(_stream(), x.data_ptr(), hi.data_ptr(),
                                      lo.data_ptr() if split else None, r, c, _as_2d_ld(x), c, 2)
    check(rc, "ps_convert_f32_to_bf16")
    return TiledBf16(hi, lo, r, c)
  shape = (c, r) if transpose else (r, c)
  hi = torch.empty(shape, dtype=torch.bfloat16, device=x.device)
  lo = torch.empty(shape, dtype=torch.bfloat16, device=x.device) if split else None
  rc = lib().ps_convert_f32_to_bf16(_stream(), x.data_ptr(), hi.data_ptr(),
                                    lo.data_ptr() if split else None, r, c, _as_2d_ld(x),
                                    shape[1], int(transpose))
  check(rc, "ps_convert_f32_to_bf16")
  return hi, lo


@_device_guarded
def gemm_bf16_grouped(items, symmetric=False):
  """items: list of ((a_hi, a_lo | None), (bt_hi, bt_lo | None), c): c [m, n] float32 =
  a [m, k] @ bt [n, k]^T on the bf16 MFMA with float32 accumulation (hi/lo pairs: three
  accumulated products).  Operands are contiguous bfloat16 tensors from to_bf16.  symmetric: every
  item is c = a a^T (bt must BE a): half of the tiles are multiplied, c is bitwise symmetric."""
  if not items:
    return
  from ._lib import GemmBf16Desc
  descs = (GemmBf16Desc * len(items))()
  dev = items[0][2].device
  for d, (a, (b_hi, b_lo), c) in zip(descs, items):
    if isinstance(a, TiledBf16):   # tile-blocked left operand (to_bf16(..., tiled=True))
      if a.layout != 1:
        raise ValueError("gemm_bf16_grouped: fragment-major operands belong to fd_filter_round")
      _require_gpu(c, "gemm_bf16_grouped")
      m, k, n = a.rows, a.cols, int(b_hi.shape[0])
      need = ((m + 127) // 128) * 128 * k
      for t in (a.hi, a.lo):
        if t is not None and (t.dtype != torch.bfloat16 or not t.is_cuda or not t.is_contiguous()
                              or t.numel() != need):
          raise ValueError("gemm_bf16_grouped: malformed tile-blocked operand")
      if int(b_hi.shape[1]) != k or tuple(c.shape) != (m, n):
        raise ValueError("gemm_bf16_grouped shape mismatch")
      for t in (b_hi, b_lo):
        if t is not None and (t.dtype != torch.bfloat16 or not t.is_cuda or t.dim() != 2 or
                              (t.shape[1] > 1 and t.stride(1) != 1)):
          raise ValueError("gemm_bf16_grouped expects row-contiguous 2-D bfloat16 device tensors")
      d.a_hi, d.a_lo = a.hi.data_ptr(), (a.lo.data_ptr() if a.lo is not None else None)
      d.b_hi, d.b_lo = b_hi.data_ptr(), (b_lo.data_ptr() if b_lo is not None else None)
      d.c, d.m, d.n, d.k = c.data_ptr(), m, n, k
      d.lda, d.ldb, d.ldc, d.a_tiled = k, _as_2d_ld(b_hi), _as_2d_ld(c), 1
      continue
    a_hi, a_lo = a
    for t in (a_hi, a_lo, b_hi, b_lo):
      if t is not None and (t.dtype != torch.bfloat16 or not t.is_cuda or t.dim() != 2 or
                            (t.shape[1] > 1 and t.stride(1) != 1)):
        raise ValueError("gemm_bf16_grouped expects row-contiguous 2-D bfloat16 device tensors")
    _require_gpu(c, "gemm_bf16_grouped")
    m, k = int(a_hi.shape[0]), int(a_hi.shape[1])
    n = int(b_hi.shape[0])
    if int(b_hi.shape[1]) != k or tuple(c.shape) != (m, n):
      raise ValueError("gemm_bf16_grouped shape mismatch")
    if (a_lo is not None and a_lo.stride() != a_hi.stride()) or (
        b_lo is not None and b_lo.stride() != b_hi.stride()):
      raise ValueError("hi and lo parts must share their layout")
    d.a_hi, d.a_lo = a_hi.data_ptr(), (a_lo.data_ptr() if a_lo is not None else None)
    d.b_hi, d.b_lo = b_hi.data_ptr(), (b_lo.data_ptr() if b_lo is not None else None)
    d.c, d.m, d.n, d.k = c.data_ptr(), m, n, k
    d.lda, d.ldb, d.ldc = _as_2d_ld(a_hi), _as_2d_ld(b_hi), _as_2d_ld(c)
    if symmetric:
      if b_hi is not a_hi or b_lo is not a_lo:
        raise ValueError("gemm_bf16_grouped(symmetric=True): both operands must be the same tensors")
      d.symmetric = 1
  L = lib()
  ws = _workspace(L.ps_gemm_bf16_grouped_workspace_bytes(descs, len(items)), dev)
  check(L.ps_gemm_bf16_grouped(_stream(), descs, len(items), ws.data_ptr(), ws.numel()),
        "ps_gemm_bf16_grouped")


@_device_guarded
def fd_filter_step(z, y, y_prev, y_next, params, step, want_bf16=True, split=True, frag=False):
  """One step of the scaled Chebyshev recurrence for the stacked iterates [B, n, b] of a
  subspace-iteration call (include/ps_api.h: ps_fd_filter_step_f32), fused with the bf16
  hi / lo split + transposition of the new iterate.  `params` = device [B, 4] float32
  {ctr, e, sigma1, degree}.  Returns (yt_hi, yt_lo) [b, B * n] bfloat16 (None, None without
  `want_bf16`; yt_lo None without `split`); frag: flat fragment-major planes of B * n * b elements
  instead (the operand layout of fd_cy_step)."""
  for t in (z, y, y_next, params):
    _require_gpu(t, "fd_filter_step")
  bsz, n, b = (int(v) for v in y.shape)
  for t in (z, y, y_next) + ((y_prev,) if y_prev is not None else ()):
    if tuple(t.shape) != (bsz, n, b) or not t.is_contiguous() or t.dtype != torch.float32:
      raise ValueError("fd_filter_step expects contiguous float32 [B, n, b] tensors")
  if tuple(params.shape) != (bsz, 4) or params.dtype != torch.float32 or not params.is_contiguous():
    raise ValueError("fd_filter_step: params must be a contiguous float32 [B, 4] tensor")
  hi = lo = None
  if want_bf16:
    shape = (bsz * n * b,) if frag else (b, bsz * n)
    hi = torch.empty(shape, dtype=torch.bfloat16, device=y.device)
    lo = torch.empty(shape, dtype=torch.bfloat16, device=y.device) if split else None
  rc = lib().ps_fd_filter_step_f32(
      _stream(), z.data_ptr(), y.data_ptr(), y_prev.data_ptr() if y_prev is not None else None,
      y_next.data_ptr(), hi.data_ptr() if hi is not None else None,
      lo.data_ptr() if lo is not None else None, params.data_ptr(), int(step), bsz, n, b,
      0 if frag else bsz * n)
  check(rc, "ps_fd_filter_step_f32")
  return hi, lo


@_device_guarded
def fd_filter_round(c16, z, bufs, params, max_degree, plain=False):
  """A whole Chebyshev filter of the subspace iteration in ONE library call
  (ps_fd_filter_round_f32): steps 1 .. max_degree of the recurrence with the grouped bf16 product
  z = C y between them.  c16[j]: the covariance of factor j as produced by to_bf16 (hi/lo pair or
  TiledBf16); z: [B, n, b] float32 holding C @ bufs[0]; bufs: three [B, n, b] float32 tensors,
  bufs[0] = the current block; params: [B, 4] from fd_round_control.  Returns the tensor of
  `bufs` that holds the filtered block."""
  from ._lib import GemmBf16Desc
  bsz, n, b = (int(v) for v in z.shape)
  for t in (z, *bufs, params):
    _require_gpu(t, "fd_filter_round")
    if not t.is_contiguous() or t.dtype != torch.float32:
      raise ValueError("fd_filter_round expects contiguous float32 tensors")
  if any(tuple(t.shape) != (bsz, n, b) for t in bufs) or tuple(params.shape) != (bsz, 4):
    raise ValueError("fd_filter_round: shape mismatch")
  dev = z.device
  ldt = bsz * n
  frag = isinstance(c16[0], TiledBf16) and c16[0].layout == 2
  if frag and (plain or any(not isinstance(a, TiledBf16) or a.layout != 2 or a.lo is None for a in c16)):
    raise ValueError("fd_filter_round: fragment-major covariances need hi/lo pairs for every factor")
  # fragment-major: two copies of the iterate planes, written and read alternately by the fused steps
  yt_shape = (2, bsz * n * b) if frag else (b, ldt)
  yt_hi = torch.empty(yt_shape, dtype=torch.bfloat16, device=dev)
  yt_lo = None if plain else torch.empty(yt_shape, dtype=torch.bfloat16, device=dev)
  descs = (GemmBf16Desc * bsz)()
  for j, d in enumerate(descs):
    a = c16[j]
    if isinstance(a, TiledBf16):
      if a.rows != n or a.cols != n:
        raise ValueError("fd_filter_round: covariance shape mismatch")
      d.a_hi, d.a_lo = a.hi.data_ptr(), (None if (plain or a.lo is None) else a.lo.data_ptr())
      d.lda, d.a_tiled = n, a.layout
    else:
      a_hi, a_lo = a
      if tuple(a_hi.shape) != (n, n):
        raise ValueError("fd_filter_round: covariance shape mismatch")
      d.a_hi, d.a_lo = a_hi.data_ptr(), (None if (plain or a_lo is None) else a_lo.data_ptr())
      d.lda, d.a_tiled = _as_2d_ld(a_hi), 0
    d.b_hi = yt_hi.data_ptr() + 2 * j * n
    d.b_lo = None if plain else yt_lo.data_ptr() + 2 * j * n
    d.c = z.data_ptr() + 4 * j * n * b
    d.m, d.n, d.k = n, b, n
    d.ldb, d.ldc = ldt, b
  L = lib()
  ws = _workspace(1024 if frag else L.ps_gemm_bf16_grouped_workspace_bytes(descs, bsz), dev)
  which = C.c_int32(-1)
  rc = L.ps_fd_filter_round_f32(
      _stream(), descs, bsz, z.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(),
      yt_hi.data_ptr(), None if plain else yt_lo.data_ptr(), params.data_ptr(), int(max_degree), n, b,
      ldt, ws.data_ptr(), ws.numel(), C.addressof(which))
  check(rc, "ps_fd_filter_round_f32")
  return bufs[which.value]


@_device_guarded
def fd_cov_update(c: torch.Tensor, grams, decay: float) -> torch.Tensor:
  """c[j] <- sym(decay * c[j] + grams[j]) in place for the stacked covariances [B, n, n] of a group of
  FD factors (ps_fd_cov_update_f32: one pass; the result is bitwise symmetric)."""
  _require_gpu(c, "fd_cov_update")
  bsz, n = int(c.shape[0]), int(c.shape[1])
  if c.dim() != 3 or c.shape[2] != n or not c.is_contiguous() or c.dtype != torch.float32 or len(grams) != bsz:
    raise ValueError("fd_cov_update expects a contiguous float32 [B, n, n] stack and B Gram matrices")
  ptrs = (C.c_void_p * bsz)()
  for j, g in enumerate(grams):
    _require_gpu(g, "fd_cov_update")
    if tuple(g.shape) != (n, n) or not g.is_contiguous() or g.dtype != torch.float32:
      raise ValueError("fd_cov_update: Gram matrices must be contiguous float32 [n, n]")
    ptrs[j] = g.data_ptr()
  check(lib().ps_fd_cov_update_f32(_stream(), c.data_ptr(), ptrs, bsz, n, float(decay)),
        "ps_fd_cov_update_f32")
  return c


def fd_frag_supported(bsz: int, n: int, b: int) -> bool:
  """Shapes the fused filter step (ps_fd_cy_step_f32) takes."""
  return bsz <= 16 and n >= 128 and n % 128 == 0 and b in (32, 64, 96)


@_device_guarded
def fd_cy_step(c16, yt, y, y_prev, y_next, nt, params, step):
  """One fused step (>= 2) of the Chebyshev filter (ps_fd_cy_step_f32): y_next = recurrence(C y, y,
  y_prev) and, with nt = (hi, lo) buffers, y_next as fragment-major bf16 planes.  c16: fragment-major
  covariances (to_bf16(..., tiled="frag")); yt = (hi, lo): the planes of y, each bsz * n * b
  bfloat16 (fd_filter_step(..., frag=True) or the previous call's nt)."""
  bsz, n, b = (int(v) for v in y.shape)
  for t in (y, y_prev, y_next, params):
    _require_gpu(t, "fd_cy_step")
    if not t.is_contiguous() or t.dtype != torch.float32:
      raise ValueError("fd_cy_step expects contiguous float32 tensors")
  if tuple(y_prev.shape) != (bsz, n, b) or tuple(y_next.shape) != (bsz, n, b) or tuple(params.shape) != (bsz, 4):
    raise ValueError("fd_cy_step: shape mismatch")
  planes = [yt[0], yt[1]] + ([nt[0], nt[1]] if nt is not None else [])
  for t in planes:
    if t.dtype != torch.bfloat16 or not t.is_cuda or not t.is_contiguous() or t.numel() != bsz * n * b:
      raise ValueError("fd_cy_step: iterate planes must be contiguous bfloat16 of bsz * n * b elements")
  ch, cl = (C.c_void_p * bsz)(), (C.c_void_p * bsz)()
  for j, a in enumerate(c16):
    if not isinstance(a, TiledBf16) or a.layout != 2 or a.lo is None or a.rows != n or a.cols != n:
      raise ValueError("fd_cy_step: covariances must be fragment-major hi/lo pairs of shape [n, n]")
    ch[j], cl[j] = a.hi.data_ptr(), a.lo.data_ptr()
  rc = lib().ps_fd_cy_step_f32(_stream(), ch, cl, bsz, yt[0].data_ptr(), yt[1].data_ptr(), y.data_ptr(),
                               y_prev.data_ptr(), y_next.data_ptr(),
                               nt[0].data_ptr() if nt is not None else None,
                               nt[1].data_ptr() if nt is not None else None, params.data_ptr(),
                               int(step), n, b)
  check(rc, "ps_fd_cy_step_f32")
  return y_next


@_device_guarded
def fd_cx6(c16, x, z, scratch=None):
  """z[j] = C_j @ x[j] to float32 accuracy on the bf16 MFMA (ps_fd_cx6_f32: three bf16 planes per
  operand, six products).  c16: covariances from to_bf16(..., tiled="frag3"); x, z: [B, n, b] float32;
  scratch: optional 3 reusable bfloat16 buffers of B * n * b elements."""
  bsz, n, b = (int(v) for v in x.shape)
  for t in (x, z):
    _require_gpu(t, "fd_cx6")
    if tuple(t.shape) != (bsz, n, b) or not t.is_contiguous() or t.dtype != torch.float32:
      raise ValueError("fd_cx6 expects contiguous float32 [B, n, b] tensors")
  if scratch is None:
    scratch = [torch.empty((bsz * n * b,), dtype=torch.bfloat16, device=x.device) for _ in range(3)]
  p0, p1, p2 = ((C.c_void_p * bsz)() for _ in range(3))
  for j, a in enumerate(c16):
    if not isinstance(a, TiledBf16) or a.layout != 2 or a.lo is None or a.lo2 is None or a.rows != n or a.cols != n:
      raise ValueError("fd_cx6: covariances must be fragment-major three-plane operands of shape [n, n]")
    p0[j], p1[j], p2[j] = a.hi.data_ptr(), a.lo.data_ptr(), a.lo2.data_ptr()
  rc = lib().ps_fd_cx6_f32(_stream(), p0, p1, p2, bsz, x.data_ptr(), z.data_ptr(), scratch[0].data_ptr(),
                           scratch[1].data_ptr(), scratch[2].data_ptr(), n, b)
  check(rc, "ps_fd_cx6_f32")
  return z


@_device_guarded
def fd_round_control(theta, res, k, n, tol, degree):
  """Per-round control of the subspace iteration on the device (ps_fd_round_control_f32).
  Returns (params [B, 4] float32, converged [B] int32, summary [4] int32 = all converged, max
  degree, min degree, plain-bf16 allowed), all on the device."""
  _require_gpu(theta, "fd_round_control")
  bsz, b = (int(v) for v in theta.shape)
  theta = theta.contiguous()
  res = res.contiguous()
  params = torch.empty((bsz, 4), dtype=torch.float32, device=theta.device)
  conv = torch.empty((bsz,), dtype=torch.int32, device=theta.device)
  summ = torch.empty((4,), dtype=torch.int32, device=theta.device)
  check(lib().ps_fd_round_control_f32(_stream(), theta.data_ptr(), res.data_ptr(), bsz, b, int(k),
                                      int(n), float(tol), int(degree), params.data_ptr(),
                                      conv.data_ptr(), summ.data_ptr()),
        "ps_fd_round_control_f32")
  return params, conv, summ


@_device_guarded
def chol_rinv_batched(gram: torch.Tensor, drop_rel: float = 1e-10, out=None) -> torch.Tensor:
  """R^-1 of the Cholesky factors G_j = R^T R of the stacked symmetric matrices [B, b, b]
  (float64 arithmetic on the device, b <= ps_chol_rinv_max_n()); directions whose pivot is
  below drop_rel * max diag(G_j) get a zero row and column."""
  _require_gpu(gram, "chol_rinv_batched")
  if gram.dim() != 3 or gram.shape[1] != gram.shape[2] or not gram.is_contiguous():
    raise ValueError("chol_rinv_batched expects a contiguous [B, b, b] tensor")
  if out is None:
    out = torch.empty_like(gram)
  elif out.shape != gram.shape or not out.is_contiguous() or out.dtype != torch.float32:
    raise ValueError("chol_rinv_batched: out must match gram")
  check(lib().ps_chol_rinv_batched_f32(_stream(), gram.data_ptr(), out.data_ptr(),
                                       int(gram.shape[1]), int(gram.shape[0]), float(drop_rel)),
        "ps_chol_rinv_batched_f32")
  return out


def tensordot_axis0(g: torch.Tensor, pc: torch.Tensor) -> torch.Tensor:
  """tensordot(g, pc, axes=[[0],[0]]) (DS:1707): contracts g's leading axis
  with pc's rows; result shape = g.shape[1:] + (pc.shape[1],)."""
  _require_gpu(g, "preconditioned_grad")
  lead = g.shape[0]
  rest = tuple(g.shape[1:])
  g2 = g.reshape(lead, -1)  # copies only if the block view is not mergeable
  out = matmul(g2, pc, transa=True)  # [prod(rest), pc.shape[1]]
  return out.reshape(rest + (pc.shape[1],))


# ---------------------------------------------------------------------------
# statistics
# ---------------------------------------------------------------------------
def gram_desc(g: torch.Tensor, axis: int, stat_in: torch.Tensor,
              stat_out: torch.Tensor):
  """Builds the ps_stats_desc for tensordot(g, g, all axes but `axis`).

  Returns (desc, keepalive).  2-D and 3-D (strided) blocks are addressed in
  place; higher ranks or exotic strides are made contiguous first and viewed
  as [outer, d, inner].
  """
  keep = [g, stat_in, stat_out]
  nd = g.dim()
  if nd == 0:
    raise ValueError("scalar gradients have no Gram matrix")

  def contiguous_3d(t):
    t = t.contiguous()
    keep.append(t)
    outer = int(np.prod(t.shape[:axis], dtype=np.int64)) if axis > 0 else 1
    inner = int(np.prod(t.shape[axis + 1:], dtype=np.int64)) if axis < nd - 1 else 1
    return t, outer, int(t.shape[axis]), inner

  d = StatsDesc()
  ok_inplace = nd <= 3 and (g.stride(nd - 1) == 1 or g.shape[nd - 1] == 1)
  if nd == 1:
    # [d]: Gram is the outer product; layout 1 with k = 1.
    gg = g.contiguous()
    keep.append(gg)
    d.g, d.layout, d.d, d.k, d.nseg, d.ld, d.seg_stride = (
        gg.data_ptr(), 1, int(gg.shape[0]), 1, 1, int(gg.shape[0]), 0)
  elif nd == 2 and ok_inplace:
    m, n = int(g.shape[0]), int(g.shape[1])
    ld = _as_2d_ld(g)
    if axis == 0:
      d.g, d.layout, d.d, d.k, d.nseg, d.ld, d.seg_stride = (
          g.data_ptr(), 0, m, n, 1, ld, 0)
    else:
      d.g, d.layout, d.d, d.k, d.nseg, d.ld, d.seg_stride = (
          g.data_ptr(), 1, n, m, 1, ld, 0)
  elif nd == 3 and ok_inplace and g.stride(2) == 1:
    b0, b1, b2 = (int(x) for x in g.shape)
    s0, s1 = int(g.stride(0)), int(g.stride(1))
    if axis == 0:    # k = (j, c): segments over j
      d.g, d.layout, d.d, d.k, d.nseg, d.ld, d.seg_stride = (
          g.data_ptr(), 0, b0, b2, b1, s0, s1)
    elif axis == 1:  # k = (a, c): segments over a
      d.g, d.layout, d.d, d.k, d.nseg, d.ld, d.seg_stride = (
          g.data_ptr(), 0, b1, b2, b0, s1, s0)
    else:            # k = (a, j): segments over a, rows j
      d.g, d.layout, d.d, d.k, d.nseg, d.ld, d.seg_stride = (
          g.data_ptr(), 1, b2, b1, b0, s1, s0)
  else:
    t, outer, dd, inner = contiguous_3d(g)
    if inner == 1:
      d.g, d.layout, d.d, d.k, d.nseg, d.ld, d.seg_stride = (
          t.data_ptr(), 1, dd, outer, 1, dd, 0)
    else:
      d.g, d.layout, d.d, d.k, d.nseg, d.ld, d.seg_stride = (
          t.data_ptr(), 0, dd, inner, outer, inner, dd * inner)
  d.stat_in, d.stat_out = stat_in.data_ptr(), stat_out.data_ptr()
  d.lds = _as_2d_ld(stat_out)
  assert _as_2d_ld(stat_in) == d.lds or stat_in.shape[0] == 1
  return d, keep


_SDESC_DT = np.dtype([("g", "u8"), ("layout", "i4"), ("d", "i4"), ("k", "i4"), ("nseg", "i4"),
                      ("ld", "i8"), ("seg_stride", "i8"), ("stat_in", "u8"),
                      ("stat_out", "u8"), ("lds", "i8")], align=True)


@_device_guarded
def stats_update_grouped(items, w1: float, w2: float):
  """items: list of (g_block, axis, stat_in, stat_out). One launch for all of them.  Row-
  contiguous 2-D blocks and contiguous vectors (the common cases) are described column-wise
  with NumPy; other ranks / strides go through gram_desc one by one."""
  if not items:
    return
  n_items = len(items)
  dev = items[0][0].device
  assert _SDESC_DT.itemsize == C.sizeof(StatsDesc)
  _same_device([t for it in items for t in (it[0], it[2], it[3])], "gram_weighted_update")
  tbl = np.zeros(n_items, _SDESC_DT)
  keep = []
  fast = [i for i, it in enumerate(items)
          if (it[0].dim() == 2 and (it[0].stride(1) == 1 or it[0].shape[1] == 1)) or
          (it[0].dim() == 1 and it[0].stride(0) == 1)]
  fast_set = set(fast)
  for i, (g, axis, sin, sout) in enumerate(items):
    if not g.is_cuda or g.dtype != torch.float32 or not sin.is_cuda:
      _require_gpu(g, "gram_weighted_update")
      _require_gpu(sin, "gram_weighted_update")
    if i not in fast_set:
      d, k = gram_desc(g, axis, sin, sout)
      keep.append(k)
      row = tbl[i]
      for name in _SDESC_DT.names:
        row[name] = getattr(d, name) or 0
  if fast:
    G = [items[i][0] for i in fast]
    SI = [items[i][2] for i in fast]
    SO = [items[i][3] for i in fast]
    # a contiguous vector block [d] is the row matrix [1, d] contracted over axis 0
    # (gram_desc: layout 1, k = 1, ld = d)
    if any(t.dim() == 1 and items[i][1] != 0 for i, t in zip(fast, G)):
      raise ValueError("axis out of range for a 1-D block")
    axis = np.array([items[i][1] if t.dim() == 2 else 1 for i, t in zip(fast, G)], np.int64)
    if np.any((axis != 0) & (axis != 1)):
      raise ValueError("axis out of range for a 2-D block")
    sg = np.array([tuple(t.shape) if t.dim() == 2 else (1, t.shape[0]) for t in G], np.int64)
    ld = _ld_2d(sg, np.array([t.stride() if t.dim() == 2 else (t.shape[0], 1) for t in G],
                             np.int64))
    so = np.array([tuple(t.shape) for t in SO], np.int64)
    lds = _ld_2d(so, np.array([t.stride() for t in SO], np.int64))
    lds_in = _ld_2d(np.array([tuple(t.shape) for t in SI], np.int64),
                    np.array([t.stride() for t in SI], np.int64))
    if np.any((lds_in != lds) & (so[:, 0] != 1)):
      raise ValueError("stat_in and stat_out must share their leading dimension")
    sub = np.zeros(len(fast), _SDESC_DT)
    sub["g"] = [t.data_ptr() for t in G]
    sub["layout"] = axis
    sub["d"] = np.where(axis == 0, sg[:, 0], sg[:, 1])
    sub["k"] = np.where(axis == 0, sg[:, 1], sg[:, 0])
    sub["nseg"] = 1
    sub["ld"] = ld
    sub["stat_in"] = [t.data_ptr() for t in SI]
    sub["stat_out"] = [t.data_ptr() for t in SO]
    sub["lds"] = lds
    tbl[fast] = sub
  descs = C.cast(tbl.ctypes.data, C.POINTER(StatsDesc))
  L = lib()
  ws = _workspace(L.ps_stats_update_grouped_workspace_bytes(descs, n_items), dev)
  rc = L.ps_stats_update_grouped_f32(_stream(), descs, n_items, float(w1),
                                     float(w2), ws.data_ptr(), ws.numel())
  check(rc, "ps_stats_update_grouped_f32")
  del keep


def gram_weighted_update(old_stats: torch.Tensor, g: torch.Tensor, axis: int,
                         w1: float, w2: float, precision=None) -> torch.Tensor:
  """DS:1440-1470: w1 * old_stats + w2 * tensordot(g, g, axes != axis)."""
  del precision  # always full float32 (exact-f32 MFMA)
  _require_gpu(g, "gram_weighted_update")
  old = old_stats.contiguous()
  out = torch.empty_like(old)
  stats_update_grouped([(g, axis, old, out)], w1, w2)
  return out


# ---------------------------------------------------------------------------
# low-rank / Frequent-Directions branch (backend interface of distributed_shampoo)
# ---------------------------------------------------------------------------
def low_rank_root(*args, **kwargs):
  from . import low_rank
  return low_rank._low_rank_root(*args, **kwargs)


def low_rank_root_batched(calls):
  """_low_rank_root for a list of keyword dicts (power iteration, eigh and the error metric
  batched; no host synchronisation)."""
  from . import low_rank
  return low_rank._low_rank_root_batched(calls)


_FD_WS = {}   # (device, bytes needed) -> workspace of the last ps_fd_update_batched_f32 call (reused across steps)


@_device_guarded
def fd_update_batched(new_grads, prev, p, rank, decay, ridge_epsilon, error_tolerance, relative_matrix_epsilon,
                      x0, input_is_factor=False, tol=1e-5, degree=12, max_outer=14):
  """ONE library call per Frequent-Directions sketch update of a group of factors (ps_fd_update_batched_f32;
  DS:1123-1290): new_grads: B contiguous float32 [d, d] Gram matrices (or factors R, input_is_factor); prev:
  [B, d, rank + 2] packed sketches; x0: [B, d, b] start block, b = fd_block_columns(rank, d).  Returns
  (packed [B, d, rank + 2], converged [B] int32 device tensor, info) or None where the library does not
  support the shape (the caller takes the step-by-step path)."""
  from ._lib import FdUpdateDesc
  L = lib()
  bsz = len(new_grads)
  d = int(new_grads[0].shape[0])
  for t in (*new_grads, prev, x0):
    _require_gpu(t, "fd_update_batched")
    if not t.is_contiguous() or t.dtype != torch.float32:
      raise ValueError("fd_update_batched expects contiguous float32 tensors")
  dev = prev.device
  u = FdUpdateDesc()
  u.batch, u.d, u.rank, u.p = bsz, d, int(rank), int(p)
  u.decay, u.ridge_epsilon, u.error_tolerance = float(decay), float(ridge_epsilon), float(error_tolerance)
  u.relative_matrix_epsilon, u.input_is_factor = int(bool(relative_matrix_epsilon)), int(bool(input_is_factor))
  u.degree, u.max_outer, u.tol = int(degree), int(max_outer), float(tol)
  need = int(L.ps_fd_update_workspace_bytes(C.byref(u)))
  if need == 0:
    return None
  b = int(L.ps_fd_block_columns(int(rank), d))
  if (tuple(prev.shape) != (bsz, d, rank + 2) or tuple(x0.shape) != (bsz, d, b) or
      any(tuple(g.shape) != (d, d) for g in new_grads)):
    raise ValueError("fd_update_batched: shape mismatch")
  key = (dev, need)
  ws = _FD_WS.get(key)
  if ws is None:
    _FD_WS.clear()           # one resident workspace (2-3 GB at 8 x 4096^2): the last shape wins
    ws = _FD_WS[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
  out = torch.empty((bsz, d, rank + 2), dtype=torch.float32, device=dev)
  conv = torch.empty((bsz,), dtype=torch.int32, device=dev)
  ptrs = (C.c_void_p * bsz)(*[g.data_ptr() for g in new_grads])
  u.new_grad = C.cast(ptrs, C.c_void_p).value
  u.prev, u.out, u.converged, u.x0 = prev.data_ptr(), out.data_ptr(), conv.data_ptr(), x0.data_ptr()
  u.workspace, u.workspace_bytes = ws.data_ptr(), ws.numel()
  info = (C.c_int32 * 4)()
  rc = L.ps_fd_update_batched_f32(_stream(), C.byref(u), C.cast(info, C.c_void_p))
  if rc == -3:               # PS_EUNSUPPORTED
    return None
  check(rc, "ps_fd_update_batched_f32")
  return out, conv, {"outer_iterations": int(info[0]), "filter_products": int(info[1]), "block": int(info[2])}


def fd_update_root(*args, **kwargs):
  from . import low_rank
  return low_rank._fd_update_root(*args, **kwargs)


def fd_update_root_batched(calls):
  """_fd_update_root for a list of keyword dicts with the eigen-step batched."""
  from . import low_rank
  return low_rank._fd_update_root_batched(calls)


# ---------------------------------------------------------------------------
# fused _transform_grad for a whole tree (DS:3496-3625)
# ---------------------------------------------------------------------------
_TDESC_DT = np.dtype([(n, "u8") for n in ("grad", "pgrad", "param", "diag_in", "diag_out",
                                          "mom_in", "mom_out", "dmom_in", "dmom_out",
                                          "upd_out")] + [("numel", "i8")], align=True)


@_device_guarded
def transform_grads_fused(items, cfg: dict):
  """items: list of dicts with contiguous float32 device tensors
  {grad, pgrad|None, param|None, diag_in|None, mom_in, dmom_in}; returns per item
  (update, new_diag|None, new_mom, new_dmom).  cfg: fields of ps_transform_config.
  The descriptor table is filled column-wise with NumPy."""
  if not items:
    return []
  n = len(items)
  dev = items[0]["grad"].device
  assert _TDESC_DT.itemsize == C.sizeof(TransformDesc)
  tbl = np.zeros(n, _TDESC_DT)
  keep = []

  def column(name, required=False):
    ptrs = np.zeros(n, np.uint64)
    col = []
    for i, it in enumerate(items):
      t = it.get(name)
      if t is None:
        if required:
          raise ValueError(f"_transform_grad: item {i} has no {name}")
        col.append(None)
        continue
      if not t.is_contiguous():
        t = t.contiguous()
      if not t.is_cuda or t.dtype != torch.float32:
        _require_gpu(t, "_transform_grad")
      col.append(t)
      ptrs[i] = t.data_ptr()
    keep.append(col)
    return ptrs, col

  tbl["grad"], grads = column("grad", required=True)
  _same_device(grads, "_transform_grad")
  for name in ("pgrad", "param", "diag_in", "mom_in", "dmom_in"):
    tbl[name], _ = column(name)
  has_diag = [it.get("diag_in") is not None for it in items]
  upd = [torch.empty_like(g) for g in grads]
  mom = [torch.empty_like(g) for g in grads]
  dmom = [torch.empty_like(g) for g in grads]
  nd = [torch.empty_like(g) if h else None for g, h in zip(grads, has_diag)]
  tbl["upd_out"] = [t.data_ptr() for t in upd]
  tbl["mom_out"] = [t.data_ptr() for t in mom]
  tbl["dmom_out"] = [t.data_ptr() for t in dmom]
  tbl["diag_out"] = [0 if t is None else t.data_ptr() for t in nd]
  tbl["numel"] = [g.numel() for g in grads]
  c = TransformConfig()
  for k, v in cfg.items():
    setattr(c, k, v)
  descs = C.cast(tbl.ctypes.data, C.POINTER(TransformDesc))
  L = lib()
  ws = _workspace(L.ps_transform_grads_workspace_bytes(descs, n), dev)
  rc = L.ps_transform_grads_f32(_stream(), descs, n, C.byref(c), ws.data_ptr(), ws.numel())
  check(rc, "ps_transform_grads_f32")
  del keep
  return list(zip(upd, nd, mom, dmom))


# ---------------------------------------------------------------------------
# quantized optimizer state (quantization_utils.py:45-113)
# ---------------------------------------------------------------------------
_QBITS = {torch.int8: 8, torch.int16: 16}


def _as_rows_cols(shape):
  """[d0, d1, ...] -> (rows, cols) with the reference's reduction axis 0 as rows."""
  if len(shape) < 1:
    raise ValueError("Input array must have a strictly positive number of dimensions.")
  rows = int(shape[0])
  cols = 1
  for d in shape[1:]:
    cols *= int(d)
  return rows, cols


_QDESC_DT = np.dtype([("fvalue", "u8"), ("codes", "u8"), ("diagonal", "u8"),
                      ("bucket_size", "u8"), ("rows", "i8"), ("cols", "i8"), ("ld", "i8"),
                      ("ldq", "i8"), ("bits", "i4"), ("extract_diagonal", "i4")], align=True)


def _contig_strides(shape):
  st, acc = [], 1
  for d in reversed(shape):
    st.append(acc)
    acc *= int(d)
  return tuple(reversed(st))


def _flat_views(shapes, dtype, device, align):
  """One allocation holding a contiguous tensor per shape (segments `align`-element
  aligned, tensors of equal shape adjacent); returns (views, flat, element offsets).
  One allocator call and one unbind per DISTINCT shape instead of an allocation per
  tensor: a parameter tree has hundreds of tensors but a handful of shapes."""
  n = len(shapes)
  groups = {}
  for i, sh in enumerate(shapes):
    groups.setdefault(tuple(sh), []).append(i)
  plan, total = [], 0
  for sh, idxs in groups.items():
    ne = 1
    for d in sh:
      ne *= int(d)
    pad = (ne + align - 1) // align * align
    plan.append((sh, idxs, ne, pad, total))
    total += pad * len(idxs)
  flat = torch.empty(total, dtype=dtype, device=device)
  views, offs = [None] * n, np.zeros(n, np.int64)
  for sh, idxs, ne, pad, base in plan:
    k = len(idxs)
    offs[idxs] = base + np.arange(k, dtype=np.int64) * pad
    if ne == 0:
      for i in idxs:
        views[i] = flat.new_empty(sh)
      continue
    seg = flat[base:base + k * pad].view(k, pad)
    if pad != ne:
      seg = seg[:, :ne]
    for i, v in zip(idxs, seg.view((k,) + sh).unbind(0)):
      views[i] = v
  return views, flat, offs


def _quant_desc_table(n):
  from ._lib import QuantDesc
  assert _QDESC_DT.itemsize == C.sizeof(QuantDesc)
  tbl = np.zeros(n, _QDESC_DT)
  return tbl, C.cast(tbl.ctypes.data, C.POINTER(QuantDesc))


@_device_guarded
def quantize_grouped(fvalues, quantized_dtype, extract_diagonal=False, out=None):
  """QuantizedValue.quantize (QU:45-95) for a list of float32 device tensors in one
  ps_quantize_f32 call.  Returns a list of (codes, diagonal | [], bucket_size);
  `out` = list of such triples of preallocated contiguous tensors to fill (e.g. views
  of an all-gather send buffer)."""
  if quantized_dtype not in _QBITS:
    raise ValueError(f"Quantized dtype {quantized_dtype} not supported.")
  if not fvalues:
    return []
  n = len(fvalues)
  dev = fvalues[0].device
  fv = []
  for f in fvalues:
    _require_gpu(f, "QuantizedValue.quantize")
    if f.dim() < 1:
      raise ValueError("Input array must have a strictly positive number of dimensions.")
    if extract_diagonal and f.dim() != 2:
      raise ValueError("Input array must be 2D to work with extract_diagonal.")
    fv.append(f if f.is_contiguous() else f.contiguous())
  shapes = [tuple(f.shape) for f in fv]
  rows = np.array([s[0] for s in shapes], np.int64)
  numel = np.array([f.numel() for f in fv], np.int64)
  cols = numel // np.maximum(rows, 1)
  tbl, descs = _quant_desc_table(n)
  tbl["fvalue"] = [f.data_ptr() for f in fv]
  if out is not None:
    for i, (codes, diag, bucket) in enumerate(out):
      if (codes.dtype != quantized_dtype or not codes.is_contiguous() or
          codes.numel() != numel[i] or bucket.numel() != cols[i] or not bucket.is_contiguous()
          or (extract_diagonal and (diag.numel() != rows[i] or not diag.is_contiguous()))):
        raise ValueError("quantize_grouped: out[i] does not match fvalues[i]")
    tbl["codes"] = [o[0].data_ptr() for o in out]
    tbl["bucket_size"] = [o[2].data_ptr() for o in out]
    if extract_diagonal:
      tbl["diagonal"] = [o[1].data_ptr() for o in out]
    outs = list(out)
  else:
    esz = 2 if quantized_dtype == torch.int16 else 1
    cviews, cflat, coff = _flat_views(shapes, quantized_dtype, dev, 16 // esz)
    bviews, bflat, boff = _flat_views([s[1:] for s in shapes], torch.float32, dev, 4)
    tbl["codes"] = cflat.data_ptr() + coff * esz
    tbl["bucket_size"] = bflat.data_ptr() + boff * 4
    if extract_diagonal:
      dviews, dflat, doff = _flat_views([(s[0],) for s in shapes], torch.float32, dev, 4)
      tbl["diagonal"] = dflat.data_ptr() + doff * 4
    else:
      dviews = [[]] * n
    outs = list(zip(cviews, dviews, bviews))
  tbl["rows"], tbl["cols"], tbl["ld"], tbl["ldq"] = rows, cols, cols, cols
  tbl["bits"] = _QBITS[quantized_dtype]
  tbl["extract_diagonal"] = int(bool(extract_diagonal))
  L = lib()
  ws = _workspace(L.ps_quantize_workspace_bytes(descs, n), dev)
  check(L.ps_quantize_f32(_stream(), descs, n, ws.data_ptr(), ws.numel()), "ps_quantize_f32")
  del fv
  return outs


class QuantizePlan:
  """quantize_grouped / dequantize_grouped on FIXED tensors (statistics or momenta updated in place,
  preallocated codes): the descriptor table and the workspace are built once, a launch is one
  C-ABI call with no per-tensor Python (the calls above spend ~0.4 ms of host time on a ViT-B
  tree, more than the kernels take).  plan = QuantizePlan(fvalues, dtype, extract, out); plan.quantize()
  fills `out` from `fvalues`, plan.dequantize() fills `fvalues` from `out`."""

  def __init__(self, fvalues, quantized_dtype, extract_diagonal, out):
    if quantized_dtype not in _QBITS:
      raise ValueError(f"Quantized dtype {quantized_dtype} not supported.")
    n = len(fvalues)
    self.n = n
    self.dev = fvalues[0].device
    for f, (codes, diag, bucket) in zip(fvalues, out):
      _require_gpu(f, "QuantizedValue.quantize")
      if not f.is_contiguous() or not codes.is_contiguous() or not bucket.is_contiguous():
        raise ValueError("QuantizePlan: contiguous tensors only")
      if codes.dtype != quantized_dtype or codes.numel() != f.numel():
        raise ValueError("QuantizePlan: out[i] does not match fvalues[i]")
      if extract_diagonal and (f.dim() != 2 or diag.numel() != f.shape[0]):
        raise ValueError("Input array must be 2D to work with extract_diagonal.")
    rows = np.array([f.shape[0] for f in fvalues], np.int64)
    numel = np.array([f.numel() for f in fvalues], np.int64)
    cols = numel // np.maximum(rows, 1)
    self.tbl, self.descs = _quant_desc_table(n)
    t = self.tbl
    t["fvalue"] = [f.data_ptr() for f in fvalues]
    t["codes"] = [o[0].data_ptr() for o in out]
    t["bucket_size"] = [o[2].data_ptr() for o in out]
    if extract_diagonal:
      t["diagonal"] = [o[1].data_ptr() for o in out]
    t["rows"], t["cols"], t["ld"], t["ldq"] = rows, cols, cols, cols
    t["bits"] = _QBITS[quantized_dtype]
    t["extract_diagonal"] = int(bool(extract_diagonal))
    L = lib()
    with torch.cuda.device(self.dev):
      self.ws = _workspace(max(L.ps_quantize_workspace_bytes(self.descs, n),
                               L.ps_dequantize_workspace_bytes(self.descs, n)), self.dev)
    self._keep = (list(fvalues), list(out))

  def quantize(self):
    with torch.cuda.device(self.dev):
      check(lib().ps_quantize_f32(_stream(), self.descs, self.n, self.ws.data_ptr(), self.ws.numel()),
            "ps_quantize_f32")

  def dequantize(self):
    with torch.cuda.device(self.dev):
      check(lib().ps_dequantize_f32(_stream(), self.descs, self.n, self.ws.data_ptr(), self.ws.numel()),
            "ps_dequantize_f32")


@_device_guarded
def dequantize_grouped(items, out=None):
  """QuantizedValue.to_float (QU:97-113) for a list of (codes, diagonal | [], bucket_size)
  in one ps_dequantize_f32 call.  Returns float32 tensors of the codes' shapes."""
  if not items:
    return []
  n = len(items)
  dev = items[0][0].device
  codes, diags, buckets = [], [], []
  extract = not (isinstance(items[0][1], list) and not items[0][1])
  for c, d, b in items:
    if not c.is_cuda or c.dtype not in _QBITS:
      raise _lib.PsError("QuantizedValue.to_float: expected int8/int16 codes on an MI355X "
                         f"device, got {c.dtype} on {c.device}; no CPU path.")
    if (not (isinstance(d, list) and not d)) != extract:
      raise ValueError("dequantize_grouped: mixed extract_diagonal in one call")
    codes.append(c if c.is_contiguous() else c.contiguous())
    buckets.append(b if b.is_contiguous() else b.contiguous())
    if extract:
      diags.append(d if d.is_contiguous() else d.contiguous())
  shapes = [tuple(c.shape) for c in codes]
  rows = np.array([s[0] for s in shapes], np.int64)
  numel = np.array([c.numel() for c in codes], np.int64)
  cols = numel // np.maximum(rows, 1)
  tbl, descs = _quant_desc_table(n)
  if out is not None:
    for f in out:
      if not f.is_contiguous():
        raise ValueError("dequantize_grouped: out tensors must be contiguous")
    outs = list(out)
    tbl["fvalue"] = [f.data_ptr() for f in outs]
  else:
    outs, fflat, foff = _flat_views(shapes, torch.float32, dev, 4)
    tbl["fvalue"] = fflat.data_ptr() + foff * 4
  tbl["codes"] = [c.data_ptr() for c in codes]
  tbl["bucket_size"] = [b.data_ptr() for b in buckets]
  if extract:
    tbl["diagonal"] = [d.data_ptr() for d in diags]
  tbl["rows"], tbl["cols"], tbl["ld"], tbl["ldq"] = rows, cols, cols, cols
  tbl["bits"] = [_QBITS[c.dtype] for c in codes]
  tbl["extract_diagonal"] = int(extract)
  L = lib()
  ws = _workspace(L.ps_dequantize_workspace_bytes(descs, n), dev)
  check(L.ps_dequantize_f32(_stream(), descs, n, ws.data_ptr(), ws.numel()),
        "ps_dequantize_f32")
  del codes, diags, buckets
  return outs
