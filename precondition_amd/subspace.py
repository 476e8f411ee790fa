"""Leading eigenpairs of symmetric PSD matrices by Chebyshev-filtered subspace iteration.

Used by the Frequent-Directions branch (config 5, DS:1123-1290), which needs only the
top rank+1 singular values / rank singular vectors of the d x d covariance update but
gets them in the reference from a full SVD of a d x (rank+d) factor (DS:1193).  Here the
heavy work is d x d @ d x b products (b = k + oversampling columns) on the fp32 MFMA GEMM
(`ps_gemm_grouped_f32`, all matrices of a call in one launch), the small b x b Gram /
Rayleigh-Ritz problems go to the batched Jacobi eigensolver (`ps_eigh_batched_f32`), and
the O(d b) recurrences are elementwise torch ops on stacked tensors.

Algorithm (Zhou & Saad's scaled Chebyshev filter inside subspace iteration):
  X <- orth(random d x b);  Rayleigh-Ritz -> theta_1 >= ... >= theta_b, X <- Ritz vectors
  repeat: X <- p(C) X with p the degree-m Chebyshev polynomial that damps [0, theta_b]
          and is scaled to 1 at theta_1;  X <- orth(X) (eigen-based Cholesky-free QR of the
          Gram matrix + one Newton-Schulz polish);  Rayleigh-Ritz;  stop when the residuals
          ||C x_i - theta_i x_i|| of the k wanted pairs are below tol * theta_1.
The caller falls back to the full eigendecomposition if this does not converge.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence, Tuple

import torch


def _K():
  from . import kernels
  return kernels


def _gemm(items):
  _K().gemm_grouped(items)


def _small_eigh_desc(mats: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
  """Batched eigh of [B, b, b] symmetric matrices, eigenvalues DESCENDING."""
  sym = 0.5 * (mats + mats.transpose(1, 2))
  es, us = _K().eigh_batched(list(sym.unbind(0)))
  e = torch.stack(es, 0).flip(1)
  u = torch.stack(us, 0).flip(2)
  return e, u


def _use_cholqr(b: int) -> bool:
  """PS_FD_CHOLQR=0 restores the eigen-based orthonormalisation."""
  import os
  from . import _lib
  return os.environ.get("PS_FD_CHOLQR", "1") != "0" and b <= _lib.lib().ps_chol_rinv_max_n()


def _fused_filter() -> bool:
  """PS_FD_FUSED_FILTER=0 restores the torch elementwise recurrence."""
  import os
  return os.environ.get("PS_FD_FUSED_FILTER", "1") != "0"


def _orthonormalize(x: torch.Tensor, tmp: torch.Tensor) -> torch.Tensor:
  """Columns of every x[j] ([B, n, b]) made orthonormal: x <- x U L^{-1/2} from the
  eigendecomposition of the Gram matrix, then one Newton-Schulz step
  x <- x (1.5 I - 0.5 x^T x).  Directions below 1e-10 of the largest Gram eigenvalue
  (a rank-deficient C) are dropped (zero columns, Ritz value 0)."""
  bsz, n, b = x.shape
  gram = torch.empty((bsz, b, b), dtype=torch.float32, device=x.device)
  _gemm([(x[j], x[j], gram[j], True, False) for j in range(bsz)])
  if _use_cholqr(b):
    # CholeskyQR: X <- X R^-1 with G = R^T R factored in float64 on the device (one small
    # launch, no b x b eigendecomposition); a pivot below 1e-10 of the largest diagonal entry
    # drops its direction like the eigen-based form below does
    m = _K().chol_rinv_batched(gram, 1e-10)
  else:
    lam, u = _small_eigh_desc(gram)
    keep = lam > 1e-10 * lam[:, :1].clamp_min(1e-30)
    scale = torch.where(keep, lam.clamp_min(1e-30).rsqrt(), torch.zeros_like(lam))
    m = u * scale[:, None, :]
  _gemm([(x[j], m[j], tmp[j], False, False) for j in range(bsz)])
  _gemm([(tmp[j], tmp[j], gram[j], True, False) for j in range(bsz)])
  eye = torch.eye(b, dtype=torch.float32, device=x.device)
  # dropped directions have a zero row/column in the Gram matrix: 1.5 I there is harmless
  polish = 1.5 * eye - 0.5 * gram
  _gemm([(tmp[j], polish[j], x[j], False, False) for j in range(bsz)])
  return x


def _rayleigh_ritz(c: Sequence[torch.Tensor], x: torch.Tensor, z: torch.Tensor,
                   tmp: torch.Tensor):
  """x orthonormal [B, n, b].  Returns (theta desc [B, b], residual norms [B, b]); x holds
  the Ritz vectors and z = C x afterwards."""
  bsz, n, b = x.shape
  _gemm([(c[j], x[j], z[j], False, False) for j in range(bsz)])
  t = torch.empty((bsz, b, b), dtype=torch.float32, device=x.device)
  _gemm([(x[j], z[j], t[j], True, False) for j in range(bsz)])
  theta, y = _small_eigh_desc(t)
  _gemm([(x[j], y[j], tmp[j], False, False) for j in range(bsz)])
  x.copy_(tmp)
  _gemm([(z[j], y[j], tmp[j], False, False) for j in range(bsz)])
  z.copy_(tmp)
  res = torch.linalg.vector_norm(z - x * theta[:, None, :], dim=1)
  return theta, res


class _Planned:
  """_orthonormalize and _rayleigh_ritz of one top_eigenpairs_batched call on persistent buffers,
  every product a kernels.GemmPlan: the eight products of an outer round are built once per call
  and launched ~6 times (PS_FD_PLANS=0 restores the per-product table building).  Same kernels,
  same operands, same order: bit-identical results."""

  def __init__(self, c, x, z, tmp, c16x6=None):
    K = _K()
    bsz, n, b = x.shape
    dev = x.device
    self.c, self.x, self.z, self.tmp = c, x, z, tmp
    self._rd = None
    self.c16x6 = c16x6      # three-plane fragment-major covariances: C x of Rayleigh-Ritz by fd_cx6
    self.x6_scratch = None
    if c16x6 is not None:
      self.x6_scratch = [torch.empty((bsz * n * b,), dtype=torch.bfloat16, device=dev) for _ in range(3)]
    mk = lambda: torch.empty((bsz, b, b), dtype=torch.float32, device=dev)
    self.gram, self.m, self.polish, self.t, self.y = mk(), mk(), mk(), mk(), mk()
    self.eye = torch.eye(b, dtype=torch.float32, device=dev)
    R = range(bsz)
    self.p_gram_x = K.GemmPlan([(x[j], x[j], self.gram[j], True, False) for j in R])
    self.p_xm = K.GemmPlan([(x[j], self.m[j], tmp[j], False, False) for j in R])
    self.p_gram_t = K.GemmPlan([(tmp[j], tmp[j], self.gram[j], True, False) for j in R])
    self.p_pol = K.GemmPlan([(tmp[j], self.polish[j], x[j], False, False) for j in R])
    self.p_cx = K.GemmPlan([(c[j], x[j], z[j], False, False) for j in R])
    self.p_xtz = K.GemmPlan([(x[j], z[j], self.t[j], True, False) for j in R])
    self.p_xy = K.GemmPlan([(x[j], self.y[j], tmp[j], False, False) for j in R])
    self.p_zy = K.GemmPlan([(z[j], self.y[j], tmp[j], False, False) for j in R])

  def round_call(self, k, tol, degree, orthonormalize=True):
    """orthonormalize (optional) + Rayleigh-Ritz + the per-round control in ONE library call
    (ps_fd_round_f32: the ~35 launches of these steps issued from C++ instead of one by one from
    Python -- the host was the limit of a round with few factors).  Returns (theta, res, params,
    converged, summary) as persistent device tensors, or None where the call does not apply."""
    from . import _lib
    L = _lib.lib()
    bsz, n, b = (int(v) for v in self.x.shape)
    if (b < 32 or not _use_cholqr(b) or b > L.ps_eigh_sorted_max_n() or
        os.environ.get("PS_FD_ROUND_LIB", "1") == "0"):
      return None
    if self._rd is None:
      dev = self.x.device
      d = _lib.FdRoundDesc()
      f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
      keep = dict(sym=f32(bsz, b, b), evals=f32(bsz, b), evecs=f32(bsz, b, b), theta=f32(bsz, b), res=f32(bsz, b),
                  params=f32(bsz, 4), converged=torch.empty((bsz,), dtype=torch.int32, device=dev),
                  summary=torch.empty((4,), dtype=torch.int32, device=dev))
      nn = (C.c_int32 * bsz)(*([b] * bsz))
      keep["ews"] = torch.empty((max(int(L.ps_eigh_root_workspace_bytes(bsz, nn)), 256),), dtype=torch.uint8, device=dev)
      for name, plan in (("gram_x", self.p_gram_x), ("xm", self.p_xm), ("gram_t", self.p_gram_t), ("pol", self.p_pol),
                         ("cx", self.p_cx), ("xtz", self.p_xtz), ("xy", self.p_xy), ("zy", self.p_zy)):
        setattr(d, name, plan._h.value)
      if self.c16x6 is not None:
        ptrs = [(C.c_void_p * bsz)() for _ in range(3)]
        for j, a in enumerate(self.c16x6):
          ptrs[0][j], ptrs[1][j], ptrs[2][j] = a.hi.data_ptr(), a.lo.data_ptr(), a.lo2.data_ptr()
        keep["ptrs"] = ptrs
        d.c0, d.c1, d.c2 = (C.cast(p, C.c_void_p).value for p in ptrs)
        d.xt0, d.xt1, d.xt2 = (t.data_ptr() for t in self.x6_scratch)
      for name, t in (("x", self.x), ("z", self.z), ("tmp", self.tmp), ("gram", self.gram), ("m", self.m),
                      ("polish", self.polish), ("t", self.t), ("y", self.y)):
        setattr(d, name, t.data_ptr())
      for name in ("sym", "evals", "evecs", "theta", "res", "params", "converged", "summary"):
        setattr(d, name, keep[name].data_ptr())
      d.eigh_workspace, d.eigh_workspace_bytes = keep["ews"].data_ptr(), keep["ews"].numel()
      d.batch, d.n, d.b = bsz, n, b
      self._rd, self._rd_keep = d, keep
    d, keep = self._rd, self._rd_keep
    d.k, d.degree, d.tol, d.orthonormalize = int(k), int(degree), float(tol), 1 if orthonormalize else 0
    with torch.cuda.device(self.x.device):
      _lib.check(L.ps_fd_round_f32(torch.cuda.current_stream().cuda_stream, C.byref(d)), "ps_fd_round_f32")
    return keep["theta"], keep["res"], keep["params"], keep["converged"], keep["summary"]

  def orthonormalize(self):
    b = self.x.shape[2]
    self.p_gram_x.launch()
    if _use_cholqr(b):
      _K().chol_rinv_batched(self.gram, 1e-10, out=self.m)
    else:
      lam, u = _small_eigh_desc(self.gram)
      keep = lam > 1e-10 * lam[:, :1].clamp_min(1e-30)
      scale = torch.where(keep, lam.clamp_min(1e-30).rsqrt(), torch.zeros_like(lam))
      self.m.copy_(u * scale[:, None, :])
    self.p_xm.launch()
    self.p_gram_t.launch()
    torch.add(self.eye * 1.5, self.gram, alpha=-0.5, out=self.polish)
    self.p_pol.launch()
    return self.x

  def rayleigh_ritz(self):
    if self.c16x6 is not None:   # z = C x on the bf16 MFMA, three planes per operand: float32 accuracy
      _K().fd_cx6(self.c16x6, self.x, self.z, self.x6_scratch)
    else:
      self.p_cx.launch()
    self.p_xtz.launch()
    theta, y = _small_eigh_desc(self.t)
    self.y.copy_(y)
    self.p_xy.launch()
    self.x.copy_(self.tmp)
    self.p_zy.launch()
    self.z.copy_(self.tmp)
    res = torch.linalg.vector_norm(self.z - self.x * theta[:, None, :], dim=1)
    return theta, res


def _filter_precision(n: int) -> str:
  """Arithmetic of the Chebyshev filter's C @ Y products (the bulk of the work; the
  Rayleigh-Ritz products and residuals are always float32 MFMA):
    "f32"    exact-f32 MFMA (v_mfma_f32_32x32x2_f32);
    "bf16x3" bf16 MFMA on hi/lo pairs of C and Y (hi*hi + lo*hi + hi*lo, float32
             accumulation): ~2^-17 operand precision at the byte traffic of float32, i.e.
             HBM-bound instead of MFMA-bound;
    "bf16"   plain bf16 operands while the residuals are above the bf16 floor, then bf16x3
             (BASELINE configs[4] names bf16 MFMA for this branch).
  Default bf16x3 where the bf16 kernel's alignment rules hold (n % 32 == 0);
  PS_FD_FILTER overrides."""
  import os
  mode = os.environ.get("PS_FD_FILTER", "bf16x3")
  if mode not in ("f32", "bf16x3", "bf16"):
    raise ValueError(f"PS_FD_FILTER must be f32, bf16x3 or bf16, got {mode!r}")
  return mode if n % 32 == 0 else "f32"


def _a_operand(c16_j, plain):
  """Left operand of a C @ Y product: the hi/lo pair (or hi alone in plain mode) of the
  covariance, row-major tuple or tile-blocked TiledBf16."""
  from .kernels import TiledBf16
  if isinstance(c16_j, TiledBf16):
    return TiledBf16(c16_j.hi, None if plain else c16_j.lo, c16_j.rows, c16_j.cols)
  return (c16_j[0], None if plain else c16_j[1])



def top_eigenpairs_batched(mats: Sequence[torch.Tensor], k: int, tol: float = 1e-5,
                           degree: int = 12, max_outer: int = 14, oversample: int = 31,
                           seed: int = 1729):
  """Leading k eigenpairs of each symmetric PSD matrix in `mats` (all n x n with the same
  n, float32, on the GPU).  Returns (evals [B, k] descending, evecs [B, n, k], converged
  [B] bool, info dict).  Deterministic (fixed-seed start block)."""
  bsz = len(mats)
  n = int(mats[0].shape[0])
  dev = mats[0].device
  degree = int(os.environ.get("PS_FD_DEGREE", degree))   # dev: cap of the per-factor filter degree
  b = min(n, ((k + oversample + 31) // 32) * 32)
  c = [m if m.is_contiguous() else m.contiguous() for m in mats]
  mode = _filter_precision(n)
  c16 = None
  if mode != "f32":  # the covariance is converted ONCE per call (hi/lo pair)
    # tile-blocked (128 x 32 tiles contiguous): the product streams whole DRAM pages of C
    tiled = n % 32 == 0 and os.environ.get("PS_FD_TILED", "1") != "0"
    # fragment-major (the kilobyte one MFMA consumes is contiguous): every filter step after the
    # first is then ONE launch -- product, recurrence and the next bf16 operand (ps_fd_cy_step_f32)
    if (tiled and mode == "bf16x3" and _fused_filter() and _K().fd_frag_supported(bsz, n, b) and
        os.environ.get("PS_FD_ROUND_CALL", "1") != "0" and os.environ.get("PS_FD_FRAG", "1") != "0"):
      # + a third plane: the float32-accurate C x of Rayleigh-Ritz runs on the bf16 MFMA as well
      tiled = "frag3" if os.environ.get("PS_FD_RR_X6", "1") != "0" else "frag"
    c16 = [_K().to_bf16(m, split=True, tiled=tiled) for m in c]

  def filter_product(y, z, plain):
    """z[j] = C_j @ y[j] for all j (y, z: [B, n, b])."""
    if c16 is None:
      _gemm([(c[j], y[j], z[j], False, False) for j in range(bsz)])
      return
    # one conversion launch for the whole stack: [B*n, b] -> [b, B*n] (k-contiguous per factor)
    yt_hi, yt_lo = _K().to_bf16(y.view(bsz * n, b), split=not plain, transpose=True)
    items = []
    for j in range(bsz):
      bt = (yt_hi[:, j * n:(j + 1) * n], None if plain else yt_lo[:, j * n:(j + 1) * n])
      a = _a_operand(c16[j], plain)
      items.append((a, bt, z[j]))
    _K().gemm_bf16_grouped(items)

  gen = torch.Generator(device=dev).manual_seed(seed)
  x = torch.randn((bsz, n, b), generator=gen, device=dev, dtype=torch.float32)
  z = torch.empty_like(x)
  tmp = torch.empty_like(x)
  x6 = c16 if (c16 is not None and tiled == "frag3") else None
  planned = _Planned(c, x, z, tmp, c16x6=x6) if os.environ.get("PS_FD_PLANS", "1") != "0" else None
  # orthonormalisation + Rayleigh-Ritz + the control of the next round in one library call where it applies
  rc = planned.round_call(k, tol, degree) if (planned is not None and _fused_filter()) else None
  lib_round = rc is not None
  params = conv_i = summ = None
  if lib_round:
    theta, res, params, conv_i, summ = rc
  elif planned is not None:
    x = planned.orthonormalize()
    theta, res = planned.rayleigh_ritz()
  else:
    x = _orthonormalize(x, tmp)
    theta, res = _rayleigh_ritz(c, x, z, tmp)
  gemms = 1
  converged = torch.zeros((bsz,), dtype=torch.bool, device=dev)
  outer = 0
  scratch = None   # two more [B, n, b] iterates of the filter, allocated once per call
  for outer in range(1, max_outer + 1):
    fused = _fused_filter()
    if fused:
      # convergence test + filter interval + per-factor degree in ONE small launch and one host
      # read (csrc/fd.hip fd_round_control_kernel; the torch form below is the same arithmetic)
      if not lib_round:
        params, conv_i, summ = _K().fd_round_control(theta, res, k, n, tol, degree)
      all_conv, max_deg, min_deg, plain_ok = summ.tolist()
      converged = conv_i.bool()
      if all_conv:
        break
      plain = mode == "bf16" and bool(plain_ok)
    else:
      top = theta[:, :1].clamp_min(1e-30)
      # pairs inside the float32 noise floor of the solver count as converged zeros
      wanted = theta[:, :k] > n * 2.4e-7 * top
      converged = ((res[:, :k] <= tol * top) | ~wanted).all(dim=1)
      if bool(converged.all()):
        break
      # Chebyshev filter: damp [0, theta_b], scaled to 1 at theta_1
      cut = theta[:, -1:].clamp_min(0.0)
      e = (0.5 * cut).clamp_min(1e-30 * top)[:, :, None]
      ctr = (0.5 * cut)[:, :, None]
      a0 = (top * (1.0 + 1e-6))[:, :, None]
      sigma1 = e / (a0 - ctr)
      sigma = sigma1
      # Per-matrix degree: the filter may amplify theta_1 over theta_k by at most ~1e2,
      # else the block collapses onto the leading directions in float32 (its Gram matrix
      # is then conditioned 1e4, which the eigen-QR + polish below still resolves).
      # T_m(x) ~ exp(m acosh x) / 2 with x = (theta - ctr) / e.
      def _acosh(v):
        v = v.clamp(1.0, 1e30)  # log v + log(1 + sqrt(1 - v^-2)): no overflow for huge v
        return torch.log(v) + torch.log1p(torch.sqrt((1.0 - 1.0 / (v * v)).clamp_min(0.0)))

      xk = theta[:, k - 1:k].clamp_min(1e-30 * top)
      spread = (_acosh((top - ctr[:, :, 0]) / e[:, :, 0]) -
                _acosh((xk - ctr[:, :, 0]) / e[:, :, 0])).clamp_min(1e-6)
      deg = torch.clamp(torch.floor(4.6 / spread), 1, degree)[:, :, None]   # [B, 1, 1]
      max_deg, min_deg = (int(v) for v in torch.stack((deg.max(), deg.min())).tolist())
      # plain bf16 operands only while the wanted residuals are far above its 2^-9 floor
      plain = mode == "bf16" and float((res[:, :k] / top).max()) > 2e-2
    if fused and c16 is not None and os.environ.get("PS_FD_ROUND_CALL", "1") != "0":
      # The whole filter in ONE library call (ps_fd_filter_round_f32): the product's task tables are
      # uploaded once and the host does nothing between the ~3 launches of a step.
      if scratch is None:
        scratch = (torch.empty_like(x), torch.empty_like(x))
      y_b = _K().fd_filter_round(c16, z, [x, scratch[0], scratch[1]], params, max_deg, plain=plain)
      gemms += max(max_deg - 1, 0)
      if y_b is not x:
        x.copy_(y_b)
    elif fused:
      # One fused launch per step (csrc/fd.hip): the recurrence for every factor + the bf16
      # hi / lo transposed copy of the new iterate that the next C @ Y product reads.
      bufs = [x, torch.empty_like(x), torch.empty_like(x)]   # y_prev, y, y_next rotate
      want16 = c16 is not None
      # z = C x is current from the Rayleigh-Ritz step
      yt = _K().fd_filter_step(z, bufs[0], None, bufs[1], params, 1, want_bf16=want16 and max_deg >= 2,
                               split=not plain)
      y_prev_b, y_b, y_next_b = bufs[0], bufs[1], bufs[2]
      for step in range(2, max_deg + 1):
        if want16:
          items = []
          for j in range(bsz):
            bt = (yt[0][:, j * n:(j + 1) * n], None if plain else yt[1][:, j * n:(j + 1) * n])
            a = _a_operand(c16[j], plain)
            items.append((a, bt, z[j]))
          _K().gemm_bf16_grouped(items)
        else:
          _gemm([(c[j], y_b[j], z[j], False, False) for j in range(bsz)])
        gemms += 1
        yt = _K().fd_filter_step(z, y_b, y_prev_b, y_next_b, params, step,
                                 want_bf16=want16 and step < max_deg, split=not plain)
        y_prev_b, y_b, y_next_b = y_b, y_next_b, y_prev_b
      if y_b is not x:
        x.copy_(y_b)
    else:
      # z = C x is current from the Rayleigh-Ritz step
      y_prev = x.clone()
      y = (z - ctr * x) * (sigma1 / e)
      for step in range(2, max_deg + 1):
        filter_product(y, z, plain)
        gemms += 1
        sigma_new = 1.0 / (2.0 / sigma1 - sigma)
        y_next = (z - ctr * y) * (2.0 * sigma_new / e) - (sigma * sigma_new) * y_prev
        if step <= min_deg:  # every matrix still filters: no masked selects over [B, n, b]
          y_prev, y, sigma = y, y_next, sigma_new
          continue
        active = deg >= step
        y_prev = torch.where(active, y, y_prev)
        y = torch.where(active, y_next, y)
        sigma = torch.where(active, sigma_new, sigma)
      x.copy_(y)
    if lib_round:
      theta, res, params, conv_i, summ = planned.round_call(k, tol, degree)
    elif planned is not None:
      x = planned.orthonormalize()
      theta, res = planned.rayleigh_ritz()
    else:
      x = _orthonormalize(x, tmp)
      theta, res = _rayleigh_ritz(c, x, z, tmp)
    gemms += 1
  else:
    top = theta[:, :1].clamp_min(1e-30)
    wanted = theta[:, :k] > n * 2.4e-7 * top
    converged = ((res[:, :k] <= tol * top) | ~wanted).all(dim=1)
  info = {"outer_iterations": outer, "big_gemms": gemms, "block": b, "filter_precision": mode,
          "max_residual_rel": float((res[:, :k] / theta[:, :1].clamp_min(1e-30)).max())}
  return theta[:, :k].contiguous(), x[:, :, :k].contiguous(), converged, info
