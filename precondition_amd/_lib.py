"""ctypes binding of libprecondition_amd.so (the C-ABI in include/ps_api.h).

The library is built in-tree by ``__graft_entry__.build()`` (or
``make -C precondition_amd/csrc``).  There is no CPU fallback: if the shared
object is missing, or a compute entry point is called without a GPU, this
module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libprecondition_amd.so")

PS_METRICS_STRIDE = 8
PS_SYMMETRY_VERIFY, PS_SYMMETRY_ASSUME, PS_SYMMETRY_GENERAL = 0, 1, 2
_SYMMETRY = {"verify": PS_SYMMETRY_VERIFY, "assume": PS_SYMMETRY_ASSUME,
             "general": PS_SYMMETRY_GENERAL}


def symmetry_code(symmetry) -> int:
  """'verify' (default: each block is tested on the device, a_ij == a_ji bit for bit, and
  blocks that fail take the full products), 'assume' (caller guarantees exact symmetry) or
  'general' (full products everywhere); see PS_SYMMETRY_* in include/ps_api.h."""
  if isinstance(symmetry, str):
    try:
      return _SYMMETRY[symmetry]
    except KeyError:
      raise ValueError(f"symmetry must be one of {sorted(_SYMMETRY)}, got {symmetry!r}")
  return int(symmetry)
(PS_M_ERROR, PS_M_ITERS, PS_M_ERROR_RATIO, PS_M_MAX_EV, PS_M_RETRIES,
 PS_M_TOTAL_ITERS, PS_M_POWER_ITERS, PS_M_AVG_STEPS) = range(8)


class PsError(RuntimeError):
  pass


class GemmDesc(C.Structure):
  """Mirror of ps_gemm_desc."""
  _fields_ = [
      ("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p),
      ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32),
      ("transa", C.c_int32), ("transb", C.c_int32),
      ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
  ]


class GemmBf16Desc(C.Structure):
  """Mirror of ps_gemm_bf16_desc."""
  _fields_ = [("a_hi", C.c_void_p), ("a_lo", C.c_void_p), ("b_hi", C.c_void_p),
              ("b_lo", C.c_void_p), ("c", C.c_void_p), ("m", C.c_int32), ("n", C.c_int32),
              ("k", C.c_int32), ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
              ("a_tiled", C.c_int32), ("symmetric", C.c_int32)]


class FdRoundDesc(C.Structure):
  """Mirror of ps_fd_round_desc."""
  _fields_ = ([(n, C.c_void_p) for n in ("gram_x", "xm", "gram_t", "pol", "cx", "xtz", "xy", "zy",
                                          "c0", "c1", "c2", "xt0", "xt1", "xt2", "x", "z", "tmp",
                                          "gram", "m", "polish", "t", "y", "sym", "evals", "evecs",
                                          "theta", "res", "eigh_workspace")] +
              [("eigh_workspace_bytes", C.c_size_t)] +
              [(n, C.c_void_p) for n in ("params", "converged", "summary")] +
              [(n, C.c_int32) for n in ("batch", "n", "b", "k", "degree", "orthonormalize")] +
              [("tol", C.c_float), ("reserved", C.c_int32)])


class FdUpdateDesc(C.Structure):
  """Mirror of ps_fd_update_desc."""
  _fields_ = ([(n, C.c_int32) for n in ("batch", "d", "rank", "p")] +
              [(n, C.c_float) for n in ("decay", "ridge_epsilon", "error_tolerance")] +
              [(n, C.c_int32) for n in ("relative_matrix_epsilon", "input_is_factor", "degree", "max_outer")] +
              [("tol", C.c_float)] +
              [(n, C.c_void_p) for n in ("new_grad", "prev", "out", "converged", "x0", "workspace")] +
              [("workspace_bytes", C.c_size_t)])


class TransformDesc(C.Structure):
  """Mirror of ps_transform_desc."""
  _fields_ = [(n, C.c_void_p) for n in ("grad", "pgrad", "param", "diag_in", "diag_out",
                                         "mom_in", "mom_out", "dmom_in", "dmom_out",
                                         "upd_out")] + [("numel", C.c_int64)]


class TransformConfig(C.Structure):
  """Mirror of ps_transform_config."""
  _fields_ = [(n, C.c_int32) for n in ("graft_type", "nesterov",
                                        "moving_average_for_momentum",
                                        "decoupled_learning_rate",
                                        "decoupled_weight_decay", "run_shampoo")] + \
             [(n, C.c_float) for n in ("beta1", "beta2_w1", "beta2_w2", "diagonal_epsilon",
                                       "weight_decay", "lr", "clip_by_scaled_gradient_norm")]


class QuantDesc(C.Structure):
  """Mirror of ps_quant_desc."""
  _fields_ = [("fvalue", C.c_void_p), ("codes", C.c_void_p), ("diagonal", C.c_void_p),
              ("bucket_size", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int64),
              ("ld", C.c_int64), ("ldq", C.c_int64), ("bits", C.c_int32),
              ("extract_diagonal", C.c_int32)]


class StatsDesc(C.Structure):
  """Mirror of ps_stats_desc."""
  _fields_ = [
      ("g", C.c_void_p),
      ("layout", C.c_int32),
      ("d", C.c_int32),
      ("k", C.c_int32),
      ("nseg", C.c_int32),
      ("ld", C.c_int64),
      ("seg_stride", C.c_int64),
      ("stat_in", C.c_void_p),
      ("stat_out", C.c_void_p),
      ("lds", C.c_int64),
  ]


PS_PRODUCTS = {"f32": 0, "bf16x6": 1, "bf16x3": 2}
PS_ACCUM = {"segmented": 0, "chain": 1}
PS_EXEC = {"staged": 0, "persistent": 1}
PS_PI = {"auto": 0, "streaming": 1, "resident": 2}
PS_EIGH = {"auto": 0, "two_sided": 1, "one_sided": 2, "tridiagonal": 3, "accurate": 4}


class PsOptions(C.Structure):
  """Mirror of ps_options (include/ps_api.h): per-call modes of the root entry points."""
  _fields_ = [
      ("struct_size", C.c_uint32),
      ("products", C.c_int32),
      ("accumulation", C.c_int32),
      ("averaged_steps", C.c_int32),
      ("iters_hint", C.c_void_p),
      ("iters_hint_stride", C.c_int32),
      ("fast_max_iters", C.c_int32),
      ("averaged_err_threshold", C.c_float),
      ("execution", C.c_int32),
      ("power_iteration", C.c_int32),
      ("pi_timeout_ms", C.c_int32),
      ("eigh_sweep_tol", C.c_float),
      ("eigh_streams", C.c_int32),
      ("eigh_solver", C.c_int32),
      ("reserved", C.c_int32 * 5),
      ("eigh_keep_max_cond", C.c_float),
      ("reserved2", C.c_int32 * 3),
  ]


def make_options(options=None):
  """dict / None -> (PsOptions, keepalive).  Keys: products ('f32' | 'bf16x6' | 'bf16x3'),
  accumulation ('segmented' | 'chain'), averaged_steps, iters_hint (host float array, one per
  block: last recompute's inverse_pth_root_iters), fast_max_iters, averaged_err_threshold,
  execution ('staged' | 'persistent'), power_iteration ('auto' | 'streaming' | 'resident'),
  pi_timeout_ms, eigh_sweep_tol, eigh_streams, eigh_solver ('auto' | 'accurate' | 'tridiagonal' | 'one_sided' |
  'two_sided'), eigh_keep_max_cond (the keep rule's lambda_max / lambda_min bound; inf = keep everything).
  Unknown keys raise."""
  import numpy as np
  o = PsOptions()
  lib().ps_options_init(C.byref(o))
  keep = []
  if not options:
    return o, keep
  opts = dict(options)
  def enum(key, table):
    v = opts.pop(key, None)
    if v is None:
      return
    if isinstance(v, str):
      if v not in table:
        raise ValueError(f"{key} must be one of {sorted(table)}, got {v!r}")
      v = table[v]
    setattr(o, key, int(v))
  enum("products", PS_PRODUCTS)
  enum("accumulation", PS_ACCUM)
  enum("execution", PS_EXEC)
  enum("power_iteration", PS_PI)
  enum("eigh_solver", PS_EIGH)
  for key in ("averaged_steps", "fast_max_iters", "pi_timeout_ms", "eigh_streams"):
    if opts.get(key) is not None:
      setattr(o, key, int(opts[key]))
    opts.pop(key, None)
  for key in ("averaged_err_threshold", "eigh_sweep_tol", "eigh_keep_max_cond"):
    if opts.get(key) is not None:
      setattr(o, key, float(opts[key]))
    opts.pop(key, None)
  hint = opts.pop("iters_hint", None)
  if hint is not None:
    if hasattr(hint, "detach"):   # a torch tensor: the hint is a HOST array (one sync if on the device)
      hint = hint.detach().to("cpu").numpy()
    h = np.ascontiguousarray(np.asarray(hint, dtype=np.float32).reshape(-1))
    keep.append(h)
    o.iters_hint = h.ctypes.data
    o.iters_hint_stride = 1
  if opts:
    raise ValueError(f"unknown option(s): {sorted(opts)}")
  return o, keep


# name -> (restype, argtypes); every symbol include/ps_api.h declares.
_SIGNATURES = {
    "ps_options_init": (None, [C.POINTER(PsOptions)]),
    "ps_newton_root_batched_opt_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p,
                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                   C.c_void_p, C.POINTER(PsOptions)]),
    "ps_eigh_root_batched_opt_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                   C.POINTER(PsOptions)]),
    "ps_power_iteration_batched_opt_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_int32, C.c_int, C.c_void_p, C.c_size_t,
                   C.POINTER(PsOptions)]),
    "ps_version": (C.c_int, []),
    "ps_error_string": (C.c_char_p, [C.c_int]),
    "ps_power_iteration_v0": (C.c_int, [C.c_int, C.c_void_p]),
    "ps_stats_update_grouped_workspace_bytes":
        (C.c_size_t, [C.POINTER(StatsDesc), C.c_int]),
    "ps_stats_update_grouped_f32":
        (C.c_int, [C.c_void_p, C.POINTER(StatsDesc), C.c_int, C.c_float,
                   C.c_float, C.c_void_p, C.c_size_t]),
    "ps_stats_update_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                   C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_float,
                   C.c_float]),
    "ps_power_iteration_workspace_bytes": (C.c_size_t, [C.c_int, C.c_void_p]),
    "ps_power_iteration_batched_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_int32, C.c_int, C.c_void_p, C.c_size_t]),
    "ps_mat_power_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ps_mat_power_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                   C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "ps_newton_root_workspace_bytes":
        (C.c_size_t, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ps_newton_averaged_steps": (C.c_int, []),
    "ps_newton_root_batched_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                   C.c_void_p]),
    "ps_newton_root_batched_maxev_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                   C.c_void_p]),
    "ps_eigh_root_workspace_bytes": (C.c_size_t, [C.c_int, C.c_void_p]),
    "ps_eigh_root_batched_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ps_gemm_grouped_workspace_bytes": (C.c_size_t, [C.POINTER(GemmDesc), C.c_int]),
    "ps_gemm_grouped_f32":
        (C.c_int, [C.c_void_p, C.POINTER(GemmDesc), C.c_int, C.c_void_p, C.c_size_t]),
    "ps_gemm_grouped_plan_create":
        (C.c_int, [C.c_void_p, C.POINTER(GemmDesc), C.c_int, C.c_void_p, C.c_size_t,
                   C.POINTER(C.c_void_p)]),
    "ps_gemm_grouped_plan_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ps_gemm_grouped_plan_destroy": (C.c_int, [C.c_void_p]),
    "ps_convert_f32_to_bf16":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                   C.c_int64, C.c_int64, C.c_int]),
    "ps_gemm_bf16_grouped_workspace_bytes": (C.c_size_t, [C.POINTER(GemmBf16Desc), C.c_int]),
    "ps_gemm_bf16_grouped":
        (C.c_int, [C.c_void_p, C.POINTER(GemmBf16Desc), C.c_int, C.c_void_p, C.c_size_t]),
    "ps_transform_grads_workspace_bytes": (C.c_size_t, [C.POINTER(TransformDesc), C.c_int]),
    "ps_transform_grads_f32":
        (C.c_int, [C.c_void_p, C.POINTER(TransformDesc), C.c_int, C.POINTER(TransformConfig),
                   C.c_void_p, C.c_size_t]),
    "ps_comm_unique_id": (C.c_int, [C.c_void_p]),
    "ps_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]),
    "ps_comm_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ps_comm_destroy": (C.c_int, [C.c_void_p]),
    "ps_comm_last_error": (C.c_char_p, []),
    "ps_quantize_workspace_bytes": (C.c_size_t, [C.POINTER(QuantDesc), C.c_int]),
    "ps_quantize_f32":
        (C.c_int, [C.c_void_p, C.POINTER(QuantDesc), C.c_int, C.c_void_p, C.c_size_t]),
    "ps_dequantize_workspace_bytes": (C.c_size_t, [C.POINTER(QuantDesc), C.c_int]),
    "ps_dequantize_f32":
        (C.c_int, [C.c_void_p, C.POINTER(QuantDesc), C.c_int, C.c_void_p, C.c_size_t]),
    "ps_eigh_batched_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ps_eigh_batched_opt_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(PsOptions)]),
    "ps_eigh_sorted_max_n": (C.c_int, []),
    "ps_fd_filter_step_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64]),
    "ps_fd_cy_step_f32":
        (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_void_p,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_int, C.c_int64, C.c_int64]),
    "ps_convert_f32_to_bf16x3_frag":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "ps_fd_cx6_f32":
        (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "ps_fd_filter_round_f32":
        (C.c_int, [C.c_void_p, C.POINTER(GemmBf16Desc), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64,
                   C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ps_fd_round_control_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                   C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ps_fd_cov_update_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_float]),
    "ps_fd_round_f32": (C.c_int, [C.c_void_p, C.POINTER(FdRoundDesc)]),
    "ps_fd_block_columns": (C.c_int, [C.c_int, C.c_int]),
    "ps_fd_update_workspace_bytes": (C.c_size_t, [C.POINTER(FdUpdateDesc)]),
    "ps_fd_update_batched_f32": (C.c_int, [C.c_void_p, C.POINTER(FdUpdateDesc), C.c_void_p]),
    "ps_chol_rinv_max_n": (C.c_int, []),
    "ps_chol_rinv_batched_f32":
        (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]),
    "ps_collective_in_flight": (C.c_int, [C.c_int]),
    "ps_power_iteration_health": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ps_power_iteration_reset_health": (C.c_int, []),
    "ps_diag_spin": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]),
    "ps_diag_mfma_mix": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "ps_diag_mfma_clock": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]),
    "ps_profile_enable": (C.c_int, [C.c_int]),
    "ps_profile_reset": (C.c_int, []),
    "ps_profile_get": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ps_gemm_f32":
        (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                   C.c_int64, C.c_int64, C.c_int64]),
}

_lib = None


def exported_symbols():
  return sorted(_SIGNATURES)


def lib():
  """Loads the shared library (once). Raises PsError if it has not been built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise PsError(
          f"{LIB_PATH} not found: build it with `python -c 'import "
          "__graft_entry__ as g; g.build()'` or `make -C precondition_amd/csrc`."
          " There is no CPU fallback.")
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
      fn = getattr(l, name)
      fn.restype = res
      fn.argtypes = args
    _lib = l
  return _lib


def check(code, what):
  if code != 0:
    msg = lib().ps_error_string(code)
    raise PsError(f"{what} failed with code {code}: "
                  f"{msg.decode() if msg else '?'}")
