// options.hip — ps_options -> psh::Options, and the library's only reader of the environment.
#include "options.h"

#include <stddef.h>
#include <stdlib.h>
#include <string.h>

namespace psh {
namespace {

// Developer overrides (A/B runs, tools/): active only under PS_DEV_ENV=1, applied on top of what
// the caller passed.  Nothing else in the library calls getenv.
void ps_dev_env_overrides(Options& o) {
  const char* gate = getenv("PS_DEV_ENV");
  if (!gate || atoi(gate) == 0) return;
  auto geti = [](const char* name, int& dst) { if (const char* e = getenv(name)) dst = atoi(e); };
  auto getf = [](const char* name, float& dst) { if (const char* e = getenv(name)) dst = (float)atof(e); };
  if (const char* e = getenv("PS_NEWTON_PRODUCTS")) {
    if (!strcmp(e, "bf16x6")) o.products = PS_PRODUCTS_BF16X6;
    else if (!strcmp(e, "bf16x3")) o.products = PS_PRODUCTS_BF16X3;
    else if (!strcmp(e, "f32")) o.products = PS_PRODUCTS_F32;
  }
  if (const char* e = getenv("PS_NEWTON_ACCUM"))
    o.accumulation = !strcmp(e, "chain") ? PS_ACCUM_CHAIN : PS_ACCUM_SEGMENTED;
  geti("PS_NEWTON_AVG_STEPS", o.averaged_steps);
  if (o.averaged_steps < 0) o.averaged_steps = 0;
  getf("PS_NEWTON_AVG_ERR", o.averaged_err_threshold);
  geti("PS_NEWTON_FAST_MAX_ITERS", o.fast_max_iters);
  if (const char* e = getenv("PS_NEWTON_PERSISTENT"))
    o.execution = atoi(e) != 0 ? PS_EXEC_PERSISTENT : PS_EXEC_STAGED;
  if (const char* e = getenv("PS_PI_RESIDENT"))
    o.power_iteration = e[0] == '0' ? PS_PI_STREAMING : PS_PI_AUTO;
  if (const char* e = getenv("PS_PI_TIMEOUT_MS")) o.pi_timeout_ms = atof(e);
  if (const char* e = getenv("PS_NEWTON_SYMMETRIC")) o.force_general = atoi(e) == 0;
  if (const char* e = getenv("PS_NEWTON_BK")) o.stage_bk = atoi(e) == 16 ? 16 : 32;
  if (const char* e = getenv("PS_NEWTON_DEEP")) o.stage_deep = o.persistent_deep = atoi(e) != 0;
  geti("PS_NEWTON_PIPE", o.pipe);
  geti("PS_NEWTON_GRID", o.grid_cap);
  geti("PS_NEWTON_GROUPS", o.newton_groups);
  geti("PS_NEWTON_WG_PER_CU", o.wg_per_cu);
  geti("PS_NEWTON_AVG_LPT", o.avg_lpt);
  geti("PS_NEWTON_STAGGER", o.stagger);
  o.newton_prof = getenv("PS_NEWTON_PROF") != nullptr;
  o.newton_trace = getenv("PS_NEWTON_TRACE");
  geti("PS_EIGH_SMALL", o.eigh_small);
  geti("PS_EIGH_SMALL_REFRESH", o.eigh_small_refresh);
  o.eigh_trace = getenv("PS_EIGH_TRACE") != nullptr;
  geti("PS_EIGH_CJ", o.eigh_cj);
  geti("PS_EIGH_CJ_REFINE", o.eigh_cj_refine);
  geti("PS_EIGH_CJ_POLISH", o.eigh_cj_polish);
  getf("PS_EIGH_CJ_TOL", o.eigh_sweep_tol);
  geti("PS_EIGH_CJ_INNER", o.eigh_cj_inner);
  getf("PS_EIGH_CJ_DONE", o.eigh_cj_done);
  geti("PS_EIGH_CJ_MAX_SWEEPS", o.eigh_cj_max_sweeps);
  geti("PS_EIGH_CJ_STATIONARY", o.eigh_cj_stationary);
  getf("PS_EIGH_CJ_ONE_BELOW", o.eigh_cj_one_below);
  geti("PS_EIGH_CJ_SORT", o.eigh_cj_sort);
  geti("PS_EIGH_CJ_STREAMS", o.eigh_streams);
  geti("PS_EIGH_CJ_UBK", o.eigh_cj_ubk);
  geti("PS_EIGH_UPDATE_X6", o.eigh_update_bf16x6);
  geti("PS_EIGH_GRAM_X6", o.eigh_gram_bf16x6);
  getf("PS_EIGH_GRAM_X3", o.eigh_gram_x3_above);
  getf("PS_EIGH_GRAM_X3_SKIP", o.eigh_gram_x3_skip);
  geti("PS_EIGH_F64_REPROJECT", o.eigh_f64_reproject);
  getf("PS_EIGH_SCALED_TOL", o.eigh_scaled_tol);
  geti("PS_EIGH_EXTRA_SWEEPS", o.eigh_extra_sweeps);
  geti("PS_EIGH_FINAL_POLISH", o.eigh_final_polish);
  geti("PS_EIGH_REFINE", o.eigh_refine);
  geti("PS_EIGH_TD", o.eigh_td);
  geti("PS_EIGH_TD_STAGE", o.eigh_td_stage);
  getf("PS_EIGH_TD_DEFL_EPS", o.eigh_td_defl_eps);
  geti("PS_EIGH_TD_STREAMS", o.eigh_td_streams);
  geti("PS_EIGH_TD_TAIL", o.eigh_td_tail);
  getf("PS_EIGH_TD_MAX_COND", o.eigh_td_max_cond);
  geti("PS_EIGH_TD_FORCE", o.eigh_td_force);
  geti("PS_EIGH_TD_ACCURATE", o.eigh_td_accurate);
  geti("PS_QUANT_FLAT", o.quant_flat);
  geti("PS_QUANT_STRIP", o.quant_strip);
  geti("PS_FD_GROUPS", o.fd_groups);
}

}  // namespace

Options resolve(const ps_options* u, bool* bad) {
  Options o;
  if (bad) *bad = false;
  if (u != nullptr) {
    // fields are read only as far as the caller's struct reaches (older builds pass a shorter one)
    ps_options c;
    memset(&c, 0, sizeof(c));
    c.averaged_steps = -1;
    c.pi_timeout_ms = -1;
    size_t sz = u->struct_size;
    if (sz < sizeof(uint32_t) || sz > sizeof(ps_options)) { if (bad) *bad = true; sz = sizeof(uint32_t); }
    memcpy(&c, u, sz);
    // a struct that reaches the marker field must carry it (ps_options_init): `ps_options o = {0}` is
    // not the defaults
    if (sz >= offsetof(ps_options, reserved) + sizeof(int32_t) && c.reserved[0] != PS_OPTIONS_MAGIC) {
      if (bad) *bad = true;
      ps_dev_env_overrides(o);
      return o;
    }
    if (c.products >= PS_PRODUCTS_F32 && c.products <= PS_PRODUCTS_BF16X3) o.products = c.products;
    else if (bad) *bad = true;
    if (c.accumulation == PS_ACCUM_SEGMENTED || c.accumulation == PS_ACCUM_CHAIN) o.accumulation = c.accumulation;
    else if (bad) *bad = true;
    if (c.averaged_steps >= 0) o.averaged_steps = c.averaged_steps;
    o.iters_hint = c.iters_hint;
    o.iters_hint_stride = c.iters_hint_stride > 0 ? c.iters_hint_stride : 1;
    if (c.fast_max_iters > 0) o.fast_max_iters = c.fast_max_iters;
    if (c.averaged_err_threshold > 0.f) o.averaged_err_threshold = c.averaged_err_threshold;
    if (c.execution == PS_EXEC_STAGED || c.execution == PS_EXEC_PERSISTENT) o.execution = c.execution;
    else if (bad) *bad = true;
    if (c.power_iteration >= PS_PI_AUTO && c.power_iteration <= PS_PI_RESIDENT) o.power_iteration = c.power_iteration;
    else if (bad) *bad = true;
    if (c.pi_timeout_ms >= 0) o.pi_timeout_ms = (double)c.pi_timeout_ms;
    if (c.eigh_sweep_tol > 0.f) o.eigh_sweep_tol = c.eigh_sweep_tol;
    if (c.eigh_streams > 0) o.eigh_streams = c.eigh_streams;
    if (c.eigh_solver == PS_EIGH_TWO_SIDED) { o.eigh_cj = 0; o.eigh_td = 0; }
    else if (c.eigh_solver == PS_EIGH_ONE_SIDED) o.eigh_td = 0;
    else if (c.eigh_solver == PS_EIGH_TRIDIAGONAL) o.eigh_td_force = 1;
    else if (c.eigh_solver == PS_EIGH_ACCURATE) o.eigh_td_accurate = 1;
    else if (c.eigh_solver != PS_EIGH_AUTO && bad) *bad = true;
    if (c.eigh_keep_max_cond > 0.f) {   // +inf included; NaN is not > 0
      o.eigh_td_max_cond = c.eigh_keep_max_cond;
      o.eigh_td_accurate = 1;           // an explicit bound applies wherever a rule could
    }
  }
  ps_dev_env_overrides(o);
  return o;
}

}  // namespace psh

extern "C" void ps_options_init(ps_options* opt) {
  if (!opt) return;
  memset(opt, 0, sizeof(*opt));
  opt->struct_size = (uint32_t)sizeof(ps_options);
  opt->products = PS_PRODUCTS_F32;
  opt->accumulation = PS_ACCUM_SEGMENTED;
  opt->averaged_steps = -1;
  opt->iters_hint_stride = 1;
  opt->pi_timeout_ms = -1;
  opt->reserved[0] = PS_OPTIONS_MAGIC;
}
