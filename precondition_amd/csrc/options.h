// options.h — every mode switch of the library, resolved once per call.
//
// psh::resolve(ps_options*) turns the caller's ps_options (include/ps_api.h; NULL = defaults)
// into the flat struct below.  The fields after `dev` have no public spelling: they are A/B
// switches of the kernels' development and keep their defaults unless the process runs with
// PS_DEV_ENV=1, in which case ps_dev_env_overrides() -- the ONLY function of the library that
// reads the environment -- applies the PS_* variables of earlier rounds on top.
#pragma once
#include "../../include/ps_api.h"

namespace psh {

struct Options {
  // ---- public (ps_options) --------------------------------------------------------------------
  int products = PS_PRODUCTS_F32;
  int accumulation = PS_ACCUM_SEGMENTED;
  int averaged_steps = 4;
  const float* iters_hint = nullptr;
  int iters_hint_stride = 1;
  int fast_max_iters = 8;
  float averaged_err_threshold = 0.f;
  int execution = PS_EXEC_STAGED;
  int power_iteration = PS_PI_AUTO;
  double pi_timeout_ms = 5000.0;
  float eigh_sweep_tol = 2e-6f;
  int eigh_streams = 0;            // 0 = by size (eigh.hip), 1 | 2
  int eigh_td = 1;                 // eigh_solver AUTO / TRIDIAGONAL: eigh_td.hip.h for 129 ... 4096 rows (PS_EIGH_TD)
  // ---- dev (environment, PS_DEV_ENV=1 only) -----------------------------------------------------
  bool force_general = false;      // PS_NEWTON_SYMMETRIC=0: full products for every block
  int stage_bk = 32;               // PS_NEWTON_BK = 16 | 32
  int stage_deep = 1;              // PS_NEWTON_DEEP
  int persistent_deep = 0;         // PS_NEWTON_DEEP (persistent execution)
  int pipe = 1;                    // PS_NEWTON_PIPE
  int grid_cap = 0;                // PS_NEWTON_GRID
  int newton_groups = 0;           // PS_NEWTON_GROUPS: stream groups of the staged execution (0 = default 2; 1 = one stream)
  int wg_per_cu = 0;               // PS_NEWTON_WG_PER_CU
  bool newton_prof = false;        // PS_NEWTON_PROF
  const char* newton_trace = nullptr;  // PS_NEWTON_TRACE=<file>
  int stagger = -1;                // PS_NEWTON_STAGGER: -1 = auto (25 us when every tile of a launch has K >= 1024), 0 = off, else (mode << 16) | microseconds (dev)
  int avg_lpt = 0;                 // PS_NEWTON_AVG_LPT: two-pass tiles first in averaged launches (measured: +1 % time)
  int eigh_small = 1;              // PS_EIGH_SMALL
  int eigh_small_refresh = 1;      // PS_EIGH_SMALL_REFRESH
  bool eigh_trace = false;         // PS_EIGH_TRACE
  int eigh_cj = 1;                 // PS_EIGH_CJ
  int eigh_cj_refine = 1;          // PS_EIGH_CJ_REFINE
  int eigh_cj_polish = 0;          // PS_EIGH_CJ_POLISH
  int eigh_cj_inner = 2;           // PS_EIGH_CJ_INNER
  float eigh_cj_done = 1e-3f;      // PS_EIGH_CJ_DONE
  int eigh_cj_max_sweeps = 24;     // PS_EIGH_CJ_MAX_SWEEPS
  int eigh_cj_stationary = 1;      // PS_EIGH_CJ_STATIONARY
  float eigh_cj_one_below = 0.1f;  // PS_EIGH_CJ_ONE_BELOW
  int eigh_cj_sort = 1;            // PS_EIGH_CJ_SORT
  int eigh_cj_ubk = 8;             // PS_EIGH_CJ_UBK
  float eigh_gram_x3_above = 0.f;    // PS_EIGH_GRAM_X3: bf16x3 Gram while the known scaled entry is above (0 = never: the default)
  float eigh_gram_x3_skip = 3e-5f;   // PS_EIGH_GRAM_X3_SKIP: pairs below this scaled entry are not rotated in bf16 sweeps
  int eigh_update_bf16x6 = 1;      // PS_EIGH_UPDATE_X6: update products on the bf16 MFMA (3-way split)
  int eigh_gram_bf16x6 = 1;        // PS_EIGH_GRAM_X6: Gram products on the bf16 MFMA, three planes, six products (float32-level)
  int eigh_f64_reproject = 1;      // PS_EIGH_F64_REPROJECT
  float eigh_scaled_tol = 1e-5f;   // PS_EIGH_SCALED_TOL
  int eigh_extra_sweeps = 4;       // PS_EIGH_EXTRA_SWEEPS
  int eigh_final_polish = 1;       // PS_EIGH_FINAL_POLISH
  int eigh_refine = 1;             // PS_EIGH_REFINE
  int eigh_td_stage = 0;           // PS_EIGH_TD_STAGE: 1 = stop after the reduction (Z_T = I; tests of the stages)
  float eigh_td_defl_eps = 1e-8f;  // PS_EIGH_TD_DEFL_EPS: deflation tolerance of the divide and conquer (x 8 ||T||)
  int eigh_td_streams = 4;         // PS_EIGH_TD_STREAMS: stream groups of the reduction
  int eigh_td_tail = 192;          // PS_EIGH_TD_TAIL: last columns of a block reduced inside LDS (0: off; <= 192)
  int fd_groups = 1;               // PS_FD_GROUPS: 2 = a Frequent-Directions update of >= 4 factors runs as two groups on two streams (dev A/B: measured +-0)
  int quant_strip = 1;             // PS_QUANT_STRIP: quantize matrices of 64 ... 4096 rows (and small tensors) from registers in one read (0 = the two-pass kernels)
  int quant_flat = 1;              // PS_QUANT_FLAT: chunks of consecutive elements for contiguous tensors (0 = 64 x 256 tiles for all)
  int eigh_td_force = 0;           // eigh_solver TRIDIAGONAL (PS_EIGH_TD_FORCE): every block keeps the fast path's result
  float eigh_td_max_cond = 1e3f;   // PS_EIGH_TD_MAX_COND / ps_options.eigh_keep_max_cond: the ACCURATE rule keeps a block's result if it is positive definite with lambda_max / lambda_min below
  int eigh_td_accurate = 0;        // eigh_solver ACCURATE: the rule above in root calls too (AUTO: root calls keep every block, plain eigh keeps the rule)
};

// PS_EINVAL-style validation is the caller's: resolve() clamps what it does not understand to the
// defaults and reports it through *bad (may be NULL).
Options resolve(const ps_options* user, bool* bad = nullptr);

}  // namespace psh
