"""Per-tree plan of the every-step calls of update() (DS:3627-3659).

Everything about a parameter tree that depends on its SHAPES only — which blocks a parameter
is cut into, where a block starts inside the (merged, contiguous) gradient, the dimensions and
leading dimensions of every Gram update and of every product of the preconditioner
application — is resolved once, into NumPy descriptor-table templates.  A step then only
fills the pointer columns (`base pointer of the parameter's gradient + byte offset of the
block`, statistics / preconditioner / output pointers) with a few vector operations and hands
the tables to the C-ABI: no block views, no per-block Python objects.  The reference gets the
same effect from tracing its Python-unrolled loops (DS:1582-1590, 1689-1708) once under jit.

Covers blocks of merged rank 1 and 2 with every axis preconditioned and no compression (the
dense mode of a transformer tree); anything else makes `TreePlan.build` return None and the
optimizer keeps its general per-block path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import kernels
from ._lib import GemmDesc, StatsDesc, check, lib


class TreePlan:

  def __init__(self):
    self.n_params = 0
    self.skipped: List[bool] = []
    self.numel: List[int] = []
    self.tshapes: List[tuple] = []
    self.n_stats_of: List[int] = []       # statistics per parameter
    self.stat_dims = np.zeros(0, np.int64)  # d of every statistic, plan order
    # statistics table
    self.stats_tbl = None
    self.st_param = None
    self.st_goff = None
    # application tables
    self.a_tbl = self.b_tbl = None
    self.a_param = self.a_goff = self.a_stat = self.a_coff = self.a_c_is_res = None
    self.b_param = self.b_stat = self.b_xoff = self.b_coff = None
    self.x_elems = 0

  @staticmethod
  def build(shapes: Sequence[tuple], pcs, skipped: Sequence[bool]) -> Optional["TreePlan"]:
    """pcs[i]: blocking.Preconditioner of parameter i (None when skipped)."""
    pl = TreePlan()
    pl.n_params = len(shapes)
    pl.skipped = list(skipped)
    st_rows, a_rows, b_rows, sa_rows, sb_rows = [], [], [], [], []
    stat_base = 0
    x_off = 0
    for i, (shape, pc) in enumerate(zip(shapes, pcs)):
      ne = 1
      for d in shape:
        ne *= int(d)
      pl.numel.append(ne)
      if pl.skipped[i]:
        pl.tshapes.append(tuple(shape))
        pl.n_stats_of.append(0)
        continue
      tshape = pc._transformed_shape
      nd = len(tshape)
      if nd not in (1, 2) or not all(pc.should_precondition_dims()) or pc._compression_rank:
        return None
      if ne == 0:
        return None
      pl.tshapes.append(tshape)
      meta = torch.empty(tshape, dtype=torch.float32, device="meta")
      row_ld = int(tshape[1]) if nd == 2 else 1
      k = 0
      for blk in pc._partitioner.partition(meta):
        off = int(blk.storage_offset()) * 4
        if nd == 1:
          d = int(blk.shape[0])
          # gram_desc: [d] -> layout 1, k = 1, ld = d
          st_rows.append((i, off, 1, d, 1, d))
          # g^T P as a one-row product: a = [d, 1] (transa), c = [1, d] inside the result
          a_rows.append((i, off, stat_base + k, d * 0 + 1, d, d, 1, d, d, True, off))
          k += 1
        else:
          m, n = int(blk.shape[0]), int(blk.shape[1])
          if blk.stride(1) != 1 or blk.stride(0) != row_ld:
            return None
          st_rows.append((i, off, 0, m, n, row_ld))   # axis 0: layout 0, d = m, k = n
          st_rows.append((i, off, 1, n, m, row_ld))   # axis 1: layout 1, d = n, k = m
          # stage A: X [n, m] = g^T P_L (transa);  stage B: Y [m, n] = X^T P_R into the result
          a_rows.append((i, off, stat_base + k, n, m, m, row_ld, m, m, False, x_off * 4))
          b_rows.append((i, stat_base + k + 1, x_off * 4, off, m, n, n, m, n, row_ld))
          # symmetric preconditioners (apply_preconditioners): X^T [m, n] = P_L g, Y = X^T P_R
          # (param, goff, stat, m, n, k, lda, ldb, ldc, xoff) / (param, stat, xoff, coff, m, n, k, lda, ldb, ldc)
          sa_rows.append((i, off, stat_base + k, m, n, m, m, row_ld, n, x_off * 4))
          sb_rows.append((i, stat_base + k + 1, x_off * 4, off, m, n, n, n, n, row_ld))
          x_off += m * n
          k += 2
      pl.n_stats_of.append(k)
      stat_base += k
    pl.x_elems = x_off
    # ---- statistics template -------------------------------------------------
    S = len(st_rows)
    tbl = np.zeros(S, kernels._SDESC_DT)
    if S:
      arr = np.array(st_rows, np.int64)
      pl.st_param, pl.st_goff = arr[:, 0].copy(), arr[:, 1].astype(np.uint64)
      tbl["layout"], tbl["d"], tbl["k"], tbl["ld"] = arr[:, 2], arr[:, 3], arr[:, 4], arr[:, 5]
      tbl["nseg"] = 1
      tbl["lds"] = arr[:, 3]
      pl.stat_dims = arr[:, 3].copy()
    pl.stats_tbl = tbl
    # ---- application templates -------------------------------------------------
    def gemm_tbl(rows):
      t = np.zeros(len(rows), kernels._GDESC_DT)
      return t
    pl.a_tbl = gemm_tbl(a_rows)
    if a_rows:
      # (param, goff, stat, m, n, k, lda, ldb, ldc, c_is_res, coff)
      pl.a_param = np.array([r[0] for r in a_rows], np.int64)
      pl.a_goff = np.array([r[1] for r in a_rows], np.uint64)
      pl.a_stat = np.array([r[2] for r in a_rows], np.int64)
      pl.a_c_is_res = np.array([r[9] for r in a_rows], bool)
      pl.a_coff = np.array([r[10] for r in a_rows], np.uint64)
      t = pl.a_tbl
      t["m"] = [r[3] for r in a_rows]
      t["n"] = [r[4] for r in a_rows]
      t["k"] = [r[5] for r in a_rows]
      t["lda"] = [r[6] for r in a_rows]
      t["ldb"] = [r[7] for r in a_rows]
      t["ldc"] = [r[8] for r in a_rows]
      t["transa"], t["transb"] = 1, 0
    pl.b_tbl = gemm_tbl(b_rows)
    if b_rows:
      # (param, stat, xoff, coff, m, n, k, lda, ldb, ldc)
      pl.b_param = np.array([r[0] for r in b_rows], np.int64)
      pl.b_stat = np.array([r[1] for r in b_rows], np.int64)
      pl.b_xoff = np.array([r[2] for r in b_rows], np.uint64)
      pl.b_coff = np.array([r[3] for r in b_rows], np.uint64)
      t = pl.b_tbl
      t["m"] = [r[4] for r in b_rows]
      t["n"] = [r[5] for r in b_rows]
      t["k"] = [r[6] for r in b_rows]
      t["lda"] = [r[7] for r in b_rows]
      t["ldb"] = [r[8] for r in b_rows]
      t["ldc"] = [r[9] for r in b_rows]
      t["transa"], t["transb"] = 1, 0
    # ---- the same application for bitwise symmetric preconditioners: P_L g instead of g^T P_L.
    # Element (i, j) of P_L g sums P[i][k] g[k][j] over k, element (j, i) of g^T P_L sums
    # g[k][j] P[k][i]: the same products in the same order when P[i][k] == P[k][i] bit for bit, so
    # X^T comes out bit-identical -- but the left operand is now k-contiguous (16-byte LDS
    # fragment reads instead of four 4-byte ones) and the stage-B product reads X^T the same way:
    # 0.72-0.77 -> 0.80-0.83 of the fp32 MFMA peak on uniform batches of ViT-B's block shapes
    # (round 4 measurement).  One-row products stay in the streaming mat-vec form.
    pl.sa_tbl = gemm_tbl(sa_rows)
    pl.sb_tbl = gemm_tbl(sb_rows)
    if sa_rows:
      pl.sa_param = np.array([r[0] for r in sa_rows], np.int64)
      pl.sa_goff = np.array([r[1] for r in sa_rows], np.uint64)
      pl.sa_stat = np.array([r[2] for r in sa_rows], np.int64)
      pl.sa_xoff = np.array([r[9] for r in sa_rows], np.uint64)
      t = pl.sa_tbl
      for col, idx in (("m", 3), ("n", 4), ("k", 5), ("lda", 6), ("ldb", 7), ("ldc", 8)):
        t[col] = [r[idx] for r in sa_rows]
      t["transa"], t["transb"] = 0, 0
      t = pl.sb_tbl
      for col, idx in (("m", 4), ("n", 5), ("k", 6), ("lda", 7), ("ldb", 8), ("ldc", 9)):
        t[col] = [r[idx] for r in sb_rows]
      t["transa"], t["transb"] = 0, 0
      # one-row products of the 1-D blocks: the rows of a_tbl whose result goes straight to the output
      pl.a_vec_rows = np.nonzero(pl.a_c_is_res)[0] if a_rows else np.zeros(0, np.int64)
    return pl

  # ---------------------------------------------------------------------------
  @staticmethod
  def _ptrs(tensors) -> np.ndarray:
    return np.fromiter((t.data_ptr() for t in tensors), np.uint64, len(tensors))

  def check_dense(self, tensors, what: str):
    """Contiguous float32 tensors on one device (the plan addresses them by pointer)."""
    dev = tensors[0].device
    for t in tensors:
      if t.dtype != torch.float32 or not t.is_cuda or t.device != dev or not t.is_contiguous():
        kernels._require_gpu(t, what)
        raise ValueError(f"{what}: the tree plan needs contiguous float32 tensors on one device")
    return dev

  def stats_update(self, grads_flat, stats_in, stats_out, w1: float, w2: float):
    """gram_weighted_update of every (block, axis) of the tree (DS:1582-1590) in one launch.
    stats_in / stats_out: flat lists in plan order (contiguous [d, d] tensors)."""
    S = len(self.stats_tbl)
    if S == 0:
      return
    dev = self.check_dense(grads_flat, "gram_weighted_update")
    if self.check_dense(list(stats_in) + list(stats_out), "gram_weighted_update") != dev:
      raise ValueError("gram_weighted_update: gradients and statistics live on different devices")
    tbl = self.stats_tbl.copy()
    gp = self._ptrs(grads_flat)
    tbl["g"] = gp[self.st_param] + self.st_goff
    tbl["stat_in"] = self._ptrs(stats_in)
    tbl["stat_out"] = self._ptrs(stats_out)
    with torch.cuda.device(dev):
      descs = C.cast(tbl.ctypes.data, C.POINTER(StatsDesc))
      L = lib()
      ws = kernels._workspace(L.ps_stats_update_grouped_workspace_bytes(descs, S), dev)
      rc = L.ps_stats_update_grouped_f32(kernels._stream(), descs, S, float(w1), float(w2),
                                         ws.data_ptr(), ws.numel())
    check(rc, "ps_stats_update_grouped_f32")

  def apply_preconditioners(self, grads_flat, precs_flat, results, symmetric_precs=False):
    """Preconditioner.preconditioned_grad (DS:1645-1708) for every planned parameter:
    two grouped launches.  precs_flat: preconditioners in plan order; results[i]: contiguous
    output of parameter i (None for skipped parameters).  symmetric_precs: the preconditioners
    are bitwise symmetric (roots of the symmetric Newton / eigh paths, identities): the products
    are issued as X^T = P_L g, Y = X^T P_R (bit-identical, faster operand layout; see build)."""
    if len(self.a_tbl) == 0:
      return
    dev = self.check_dense(grads_flat, "preconditioned_grad")
    self.check_dense([p for p in precs_flat] + [r for r in results if r is not None],
                     "preconditioned_grad")
    x = torch.empty(max(self.x_elems, 1), dtype=torch.float32, device=dev)
    xp = np.uint64(x.data_ptr())
    gp = self._ptrs(grads_flat)
    pp = self._ptrs(precs_flat)
    rp = np.fromiter((0 if r is None else r.data_ptr() for r in results), np.uint64,
                     len(results))
    ta = self.a_tbl.copy()
    ta["a"] = gp[self.a_param] + self.a_goff
    ta["b"] = pp[self.a_stat]
    ta["c"] = np.where(self.a_c_is_res, rp[self.a_param], xp) + self.a_coff
    sym = (symmetric_precs and len(getattr(self, "sa_tbl", ())) > 0 and
           os.environ.get("PS_APPLY_SYM", "1") != "0")
    if sym:
      vec = ta[self.a_vec_rows]
      sa = self.sa_tbl.copy()
      sa["a"] = pp[self.sa_stat]
      sa["b"] = gp[self.sa_param] + self.sa_goff
      sa["c"] = xp + self.sa_xoff
      ta = np.concatenate([sa, vec]) if len(vec) else sa
    L = lib()
    with torch.cuda.device(dev):
      for tbl in (ta, None):
        if tbl is None:
          if len(self.b_tbl) == 0:
            break
          if sym:
            tbl = self.sb_tbl.copy()
          else:
            tbl = self.b_tbl.copy()
          tbl["a"] = xp + self.b_xoff
          tbl["b"] = pp[self.b_stat]
          tbl["c"] = rp[self.b_param] + self.b_coff
        n = len(tbl)
        descs = C.cast(tbl.ctypes.data, C.POINTER(GemmDesc))
        ws = kernels._workspace(L.ps_gemm_grouped_workspace_bytes(descs, n), dev)
        rc = L.ps_gemm_grouped_f32(kernels._stream(), descs, n, ws.data_ptr(), ws.numel())
        check(rc, "ps_gemm_grouped_f32")
    del x  # the caching allocator keeps it alive until the stream has consumed it


class DonatedStep:
  """The every-step calls of update() on a DONATED state (distributed_shampoo(donate_state=True)):
  the counterpart of the reference's jit-compiled update_fn (DS:3627-3659), which does no per-step
  host work beyond the dispatch.

  The state tensors (statistics, diagonal statistics, both momenta) are updated IN PLACE, the
  preconditioned gradients, the intermediate X of the two-sided application and the updates live
  in buffers this object owns, so every pointer column of the four descriptor tables (Gram
  update, application stage A and B, _transform_grad) is filled ONCE per binding; a step patches
  only the gradient (and, with weight decay, parameter) columns when those pointers moved, and
  issues the same seven launches as the functional path.  No torch.empty, no per-tensor Python
  objects, no pytree rebuilt: the caller gets the same state objects back.

  Bound to one list of ParameterStats objects (identity is checked each step: a state the caller
  rebuilt, or the one a recompute step returned, is simply bound again)."""

  def __init__(self, plan: TreePlan, has_diag: bool, use_params: bool):
    self.plan = plan
    self.has_diag = has_diag
    self.use_params = use_params
    self.objs = None
    self.gp = None
    self.pp_params = None

  def bound_to(self, stats_flat) -> bool:
    objs = self.objs
    return (objs is not None and len(objs) == len(stats_flat) and
            all(a is b for a, b in zip(objs, stats_flat)))

  def bind(self, stats_flat, grads_flat, params_flat, symmetric_precs: bool):
    pl = self.plan
    dev = pl.check_dense(grads_flat, "update (donated state)")
    stats = [s for st in stats_flat for s in st.statistics]
    precs = [p for st in stats_flat for p in st.preconditioners]
    moms = [st.momentum.quantized for st in stats_flat]
    dmoms = [st.diagonal_momentum.quantized for st in stats_flat]
    diags = [st.diagonal_statistics.quantized for st in stats_flat] if self.has_diag else []
    if len(stats) != len(pl.stat_dims) or len(precs) != len(pl.stat_dims):
      return False
    pl.check_dense(stats + precs + moms + dmoms + diags, "update (donated state)")
    n = pl.n_params
    self.dev = dev
    # buffers owned by the step (valid until the next update call)
    if getattr(self, "upd", None) is None:
      self.upd = [torch.empty_like(g) for g in grads_flat]
      self.pg = [None if sk else torch.empty_like(g) for g, sk in zip(grads_flat, pl.skipped)]
      self.x = torch.empty(max(pl.x_elems, 1), dtype=torch.float32, device=dev)
    sp, prp = pl._ptrs(stats), pl._ptrs(precs)
    rp = np.fromiter((0 if r is None else r.data_ptr() for r in self.pg), np.uint64, n)
    xp = np.uint64(self.x.data_ptr())
    # ---- Gram update, in place ----
    st = pl.stats_tbl.copy()
    st["stat_in"] = sp
    st["stat_out"] = sp
    self.st_tbl = st
    # ---- application ----
    ta = pl.a_tbl.copy()
    ta["b"] = prp[pl.a_stat] if len(ta) else prp[:0]
    ta["c"] = (np.where(pl.a_c_is_res, rp[pl.a_param], xp) + pl.a_coff) if len(ta) else prp[:0]
    self.sym = bool(symmetric_precs and len(getattr(pl, "sa_tbl", ())) > 0 and
                    os.environ.get("PS_APPLY_SYM", "1") != "0")
    if self.sym:
      vec = ta[pl.a_vec_rows]
      sa = pl.sa_tbl.copy()
      sa["a"] = prp[pl.sa_stat]
      sa["c"] = xp + pl.sa_xoff
      self.n_sa = len(sa)
      self.ta = np.concatenate([sa, vec]) if len(vec) else sa
      tb = pl.sb_tbl.copy()
    else:
      self.ta = ta
      tb = pl.b_tbl.copy()
    if len(tb):
      tb["a"] = xp + pl.b_xoff
      tb["b"] = prp[pl.b_stat]
      tb["c"] = rp[pl.b_param] + pl.b_coff
    self.tb = tb
    # ---- _transform_grad, in place ----
    tt = np.zeros(n, kernels._TDESC_DT)
    tt["pgrad"] = rp
    mp, dp = pl._ptrs(moms), pl._ptrs(dmoms)
    tt["mom_in"] = mp; tt["mom_out"] = mp
    tt["dmom_in"] = dp; tt["dmom_out"] = dp
    if self.has_diag:
      gp_ = pl._ptrs(diags)
      tt["diag_in"] = gp_; tt["diag_out"] = gp_
    tt["upd_out"] = pl._ptrs(self.upd)
    tt["numel"] = pl.numel
    self.tt = tt
    L = lib()
    with torch.cuda.device(dev):
      def ws_for(nbytes):
        return torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=dev)
      self.ws_stats = ws_for(L.ps_stats_update_grouped_workspace_bytes(
          C.cast(self.st_tbl.ctypes.data, C.POINTER(StatsDesc)), len(self.st_tbl))) if len(self.st_tbl) else None
      self.ws_a = ws_for(L.ps_gemm_grouped_workspace_bytes(
          C.cast(self.ta.ctypes.data, C.POINTER(GemmDesc)), len(self.ta))) if len(self.ta) else None
      self.ws_b = ws_for(L.ps_gemm_grouped_workspace_bytes(
          C.cast(self.tb.ctypes.data, C.POINTER(GemmDesc)), len(self.tb))) if len(self.tb) else None
      tt_probe = self.tt.copy()
      tt_probe["grad"] = tt_probe["upd_out"]   # sizes only
      self.ws_t = ws_for(L.ps_transform_grads_workspace_bytes(
          C.cast(tt_probe.ctypes.data, C.POINTER(kernels.TransformDesc)), n))
    self.objs = list(stats_flat)
    self.gp = None
    self.pp_params = None
    return True

  def qualifies(self, grads_flat, params_flat) -> bool:
    """Checked on EVERY step, before the pointer comparison: the raw-pointer tables read a gradient as a
    contiguous float32 tensor of the bound shape.  A tensor with the same data_ptr and shape but other
    strides (a `.t()` view of a square weight, an expanded tensor) or another dtype must not reach them;
    such a step takes the functional path (which copies / converts), it does not raise."""
    for g, u in zip(grads_flat, self.upd):
      if (g.dtype != torch.float32 or not g.is_contiguous() or g.device != self.dev or
          g.shape != u.shape):
        return False
    if self.use_params:
      for p_, u in zip(params_flat, self.upd):
        if (p_.dtype != torch.float32 or not p_.is_contiguous() or p_.device != self.dev or
            p_.shape != u.shape):
          return False
    return True

  def _patch_grads(self, grads_flat):
    pl = self.plan
    gp = pl._ptrs(grads_flat)
    if self.gp is not None and np.array_equal(gp, self.gp):
      return
    self.gp = gp
    if len(self.st_tbl):
      self.st_tbl["g"] = gp[pl.st_param] + pl.st_goff
    if self.sym:
      self.ta["b"][:self.n_sa] = gp[pl.sa_param] + pl.sa_goff
      if len(self.ta) > self.n_sa:
        vr = pl.a_vec_rows
        self.ta["a"][self.n_sa:] = gp[pl.a_param[vr]] + pl.a_goff[vr]
    elif len(self.ta):
      self.ta["a"] = gp[pl.a_param] + pl.a_goff
    self.tt["grad"] = gp

  def step(self, grads_flat, params_flat, cfg, do_stats: bool, w1: float, w2: float):
    """Enqueues the step; returns the update tensors (owned by this object)."""
    self._patch_grads(grads_flat)    # (the caller checked qualifies() for this step)
    if self.use_params:
      pp = self.plan._ptrs(params_flat)
      if self.pp_params is None or not np.array_equal(pp, self.pp_params):
        self.pp_params = pp
        self.tt["param"] = pp
    L = lib()
    stream = kernels._stream()
    with torch.cuda.device(self.dev):
      if do_stats and len(self.st_tbl):
        rc = L.ps_stats_update_grouped_f32(stream, C.cast(self.st_tbl.ctypes.data, C.POINTER(StatsDesc)),
                                           len(self.st_tbl), float(w1), float(w2),
                                           self.ws_stats.data_ptr(), self.ws_stats.numel())
        check(rc, "ps_stats_update_grouped_f32")
      for tbl, ws in ((self.ta, self.ws_a), (self.tb, self.ws_b)):
        if len(tbl):
          rc = L.ps_gemm_grouped_f32(stream, C.cast(tbl.ctypes.data, C.POINTER(GemmDesc)), len(tbl),
                                     ws.data_ptr(), ws.numel())
          check(rc, "ps_gemm_grouped_f32")
      c = kernels.TransformConfig()
      for k, v in cfg.items():
        setattr(c, k, v)
      rc = L.ps_transform_grads_f32(stream, C.cast(self.tt.ctypes.data, C.POINTER(kernels.TransformDesc)),
                                    self.plan.n_params, C.byref(c), self.ws_t.data_ptr(), self.ws_t.numel())
      check(rc, "ps_transform_grads_f32")
    return self.upd
