// fd_update.hip — ps_fd_update_batched_f32: ONE library call per Frequent-Directions sketch update of a group of
// equally shaped factors (BASELINE configs[4]; reference: _fd_update_root DS:1123-1290, one call per factor under
// vmap, DS:2732-2738).  SURVEY.md section 8(b) proposed this entry point; until round 6 the Python host issued the
// steps below one by one (precondition_amd/low_rank.py _fd_update_root_group + subspace.py top_eigenpairs_batched):
//
//   prepare   W_j = sketch_j * sqrt(eigs_j + ridge_j)        (fd_prep_kernel; DS:1160-1172)
//             C_j = sym(decay W_j W_j^T + Gram_j)            (ps_gemm_grouped_f32, ps_fd_cov_update_f32; DS:1174-1193)
//             three fragment-major bf16 planes of C_j        (ps_convert_f32_to_bf16x3_frag)
//   iterate   block subspace iteration for the leading rank + 1 eigenpairs of C_j (= u, s^2 of the SVD at DS:1193):
//             per outer round ps_fd_round_f32 (CholeskyQR + Rayleigh-Ritz + control), ONE 16-byte host read, and
//             ps_fd_filter_round_f32 (Chebyshev filter on the bf16 MFMA); at most max_outer rounds
//   finish    deflation by the cutoff singular value, tail update, sanity masks, inverted eigenvalues, packing into
//             the [d, rank + 2] layout of DS:555-592   (fd_finish_kernel; DS:1196-1290)
//
// The arithmetic of prepare / finish is the float32 expression sequence of the reference (and of the torch
// statements it replaces) term for term; only the column norms are summed in another order.  The iteration is the
// same sequence of library calls on the same buffers as before: with the caller's start block x0 the eigenpairs are
// bit-identical to the Python-driven path.  Shapes outside the fused kernels' domain return PS_EUNSUPPORTED (the
// Python host then takes its general path); a factor whose iteration does not converge is flagged in `converged`.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "gemm_core.hip.h"
#include "options.h"

namespace psk {

// W = sketch * sqrt(eigs + ridge) for factor j: prev [d][r + 2] packed (DS:555-592): sketch = columns 0 .. r-1,
// deflated eigenvalues = last column, rows d-r .. d-1, tail = last column row 1.
__global__ __launch_bounds__(256) void fd_prep_kernel(const float* prev, float* weighted, int d, int r,
                                                      float ridge_eps, float err_tol, int relative) {
  const int j = blockIdx.y;
  const float* pj = prev + (int64_t)j * d * (r + 2);
  float* wj = weighted + (int64_t)j * d * r;
  const float fwd0 = pj[(int64_t)(d - r) * (r + 2) + r + 1];
  const float ridge = relative ? __fmul_rn(ridge_eps, fmaxf(fwd0, err_tol)) : __fmul_rn(ridge_eps, fmaxf(1.f, err_tol));
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)d * r; e += (int64_t)gridDim.x * 256) {
    const int row = (int)(e / r), c = (int)(e % r);
    const float ev = pj[(int64_t)(d - r + c) * (r + 2) + r + 1];
    wj[e] = __fmul_rn(pj[(int64_t)row * (r + 2) + c], sqrtf(__fadd_rn(ev, ridge)));
  }
}

// Partial sums of squares of the Ritz vectors' columns: slice s of FD_NP row slices of factor j ->
// part[(j * FD_NP + s) * r + c].  (As ONE workgroup per factor walking column after column this was 0.7 ms of a
// 9 ms one-factor update: 64 strided passes over 4096 rows.)  Lane = column, so a row is one contiguous run; the row
// groups of a workgroup are added in a fixed order, the slices in a fixed order by fd_finish_kernel.
constexpr int FD_NP = 32;   // row slices of the column norms
constexpr int FD_NS = 16;   // row slices of the packing
__global__ __launch_bounds__(256) void fd_colnorm_part_kernel(const float* x, float* part, int d, int b, int r) {
  __shared__ float red[256];
  const int s = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
  const float* xj = x + (int64_t)j * d * b;
  const int rows_per = (d + FD_NP - 1) / FD_NP;
  const int row0 = s * rows_per, row1 = row0 + rows_per < d ? row0 + rows_per : d;
  for (int c0 = 0; c0 < r; c0 += 256) {
    const int rc = r - c0 < 256 ? r - c0 : 256;          // columns of this pass
    const int cw = (rc + 63) & ~63;                       // lanes per row group
    const int rg_n = 256 / cw, c = tid % cw, rg = tid / cw;
    float ss = 0.f;
    if (rg < rg_n && c < rc) {
      int row = row0 + rg;
      for (; row + 3 * rg_n < row1; row += 4 * rg_n) {
        const float v0 = xj[(int64_t)row * b + c0 + c], v1 = xj[(int64_t)(row + rg_n) * b + c0 + c];
        const float v2 = xj[(int64_t)(row + 2 * rg_n) * b + c0 + c], v3 = xj[(int64_t)(row + 3 * rg_n) * b + c0 + c];
        ss = __fmaf_rn(v0, v0, ss); ss = __fmaf_rn(v1, v1, ss); ss = __fmaf_rn(v2, v2, ss); ss = __fmaf_rn(v3, v3, ss);
      }
      for (; row < row1; row += rg_n) { const float v = xj[(int64_t)row * b + c0 + c]; ss = __fmaf_rn(v, v, ss); }
    }
    red[tid] = ss;
    __syncthreads();
    if (tid < rc) {
      float t = red[tid];
      for (int g = 1; g < rg_n; ++g) t += red[g * cw + tid];
      part[((int64_t)j * FD_NP + s) * r + c0 + tid] = t;
    }
    __syncthreads();
  }
}

// DS:1196-1290 for factor j: theta [b] descending Ritz values, x [d][b] Ritz vectors (the first r columns are used),
// prev as above, part = fd_colnorm_part_kernel's sums -> out [d][r + 2].  grid (FD_NS, B): every workgroup evaluates
// the O(r) scalars (same arithmetic, same values) and packs its slice of rows.
__global__ __launch_bounds__(256) void fd_finish_kernel(const float* theta, const float* x, const float* prev,
                                                        const float* part, float* out, int d, int b, int r, int p,
                                                        float decay) {
  extern __shared__ float sm[];   // e[r + 1], deflated[r], scale[r], inv[r], red[r]
  const int j = blockIdx.y, tid = threadIdx.x;
  float* e = sm;
  float* defl = e + (r + 1);
  float* scale = defl + r;
  float* inv = scale + r;
  float* red = inv + r;
  const float* th = theta + (int64_t)j * b;
  const float* xj = x + (int64_t)j * d * b;
  const float* pj = prev + (int64_t)j * d * (r + 2);
  float* oj = out + (int64_t)j * d * (r + 2);
  __shared__ float s_scalars[4];   // new_const, new_tail, has_zeros, tail (decayed)
  // column norms: the slices' partial sums in slice order
  for (int c = tid; c < r; c += 256) {
    float ss = 0.f;
    for (int s = 0; s < FD_NP; ++s) ss += part[((int64_t)j * FD_NP + s) * r + c];
    red[c] = ss;
  }
  if (tid == 0) {
    float emax = th[0];
    for (int c = 1; c <= r; ++c) emax = fmaxf(emax, th[c]);
    const float noise = __fmul_rn((float)((double)d * 1.2e-7), fmaxf(emax, 0.f));
    for (int c = 0; c <= r; ++c) {
      const float v = th[c] <= noise ? 0.f : th[c];
      e[c] = sqrtf(fmaxf(v, 0.f));                      // singular values s
    }
    const float cutoff = e[r];
    const float rho = __fmul_rn(cutoff, cutoff);
    const float tail = __fmul_rn(pj[(int64_t)1 * (r + 2) + r + 1], decay);
    float new_tail = __fadd_rn(tail, rho);
    const float alpha = (float)(-1.0 / (double)p);
    const float new_const = new_tail <= 0.f ? 0.f : powf(new_tail, alpha);
    new_tail = new_tail <= 0.f ? 0.f : new_tail;
    for (int c = 0; c < r; ++c) {
      const float dv = __fmul_rn(__fsub_rn(e[c], cutoff), __fadd_rn(e[c], cutoff));
      defl[c] = dv <= 0.f ? 0.f : dv;
    }
    s_scalars[0] = new_const; s_scalars[1] = new_tail; s_scalars[3] = tail;
  }
  __syncthreads();
  if (tid == 0) {
    const float alpha = (float)(-1.0 / (double)p);
    const float tail = s_scalars[3];
    bool zeros = s_scalars[1] <= 0.f;
    for (int c = 0; c < r; ++c) {
      // (columns with a zero deflated eigenvalue are zeroed before the norms are taken: norm 0, not safe)
      const float nrm = defl[c] > 0.f ? sqrtf(red[c]) : 0.f;
      const bool safe = 0.99f <= nrm && nrm <= 1.01f;
      const float dv = safe ? defl[c] : 0.f;            // deflated *= safe
      // eigvecs = eigvecs * (deflated > 0) * safe / where(safe, norms, 1): keep the norm (0 = dropped column)
      scale[c] = (defl[c] > 0.f && safe) ? nrm : 0.f;
      float up = __fadd_rn(__fmul_rn(e[c], e[c]), tail);
      up = dv > 0.f ? up : 0.f;
      up = up <= 0.f ? 0.f : up;
      inv[c] = up <= 0.f ? 0.f : powf(up, alpha);
      defl[c] = dv;
      zeros |= dv <= 0.f;
    }
    s_scalars[2] = zeros ? 1.f : 0.f;
  }
  __syncthreads();
  const int pd = r + 2;
  const int rows_per = (d + FD_NS - 1) / FD_NS;
  const int row0 = blockIdx.x * rows_per, row1 = row0 + rows_per < d ? row0 + rows_per : d;
  for (int64_t el = (int64_t)row0 * pd + tid; el < (int64_t)row1 * pd; el += 256) {
    const int row = (int)(el / pd), c = (int)(el % pd);
    float v = 0.f;
    if (c < r) {
      const float nrm = scale[c];
      v = nrm != 0.f ? __fdiv_rn(xj[(int64_t)row * b + c], nrm) : 0.f;
    } else if (c == r) {
      if (row < r) v = inv[row];
      if (row == d - 1) v = s_scalars[2];
    } else {
      if (row == 0) v = s_scalars[0];
      else if (row == 1) v = s_scalars[1];
      if (row >= d - r) v = defl[row - (d - r)];
    }
    oj[el] = v;
  }
}

__global__ void fd_fill_int_kernel(int32_t* p, int n, int32_t v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

namespace {
inline int fd_block_columns(int rank, int d) {
  const int b = ((rank + 1 + 31 + 31) / 32) * 32;
  return b < d ? b : d;
}

struct FdCarve {
  float *c, *gram, *weighted, *x, *z, *tmp, *s0, *s1;
  float *g, *m, *polish, *t, *y, *sym, *evecs, *evals, *theta, *res, *params;
  int32_t *conv, *summary;
  void *planes[3], *xt[3], *yt_hi, *yt_lo, *eigh_ws, *plan_ws[10], *cy_ws;
  size_t eigh_ws_bytes, plan_ws_bytes[10];
};

// sizes only depend on (B, d, r, factor input): the plan workspaces are bounded by their descriptors' shapes
size_t fd_carve(psh::Arena& ar, FdCarve* o, int B, int d, int r, int b, bool factor_input,
                const size_t plan_bytes[10], size_t eigh_bytes) {
  const size_t dd = (size_t)B * d * d, db = (size_t)B * d * b, bb = (size_t)B * b * b;
  FdCarve t;
  t.c = ar.take<float>(dd);
  t.gram = factor_input ? ar.take<float>(dd) : nullptr;
  t.weighted = ar.take<float>((size_t)B * d * r);
  t.x = ar.take<float>(db); t.z = ar.take<float>(db); t.tmp = ar.take<float>(db);
  t.s0 = ar.take<float>(db); t.s1 = ar.take<float>(db);
  t.g = ar.take<float>(bb); t.m = ar.take<float>(bb); t.polish = ar.take<float>(bb); t.t = ar.take<float>(bb);
  t.y = ar.take<float>(bb); t.sym = ar.take<float>(bb); t.evecs = ar.take<float>(bb);
  t.evals = ar.take<float>((size_t)B * b); t.theta = ar.take<float>((size_t)B * b); t.res = ar.take<float>((size_t)B * b);
  t.params = ar.take<float>((size_t)B * 4);
  t.conv = ar.take<int32_t>(B); t.summary = ar.take<int32_t>(4);
  for (int k = 0; k < 3; ++k) t.planes[k] = ar.take<uint16_t>(dd);
  for (int k = 0; k < 3; ++k) t.xt[k] = ar.take<uint16_t>(db);
  t.yt_hi = ar.take<uint16_t>(2 * db); t.yt_lo = ar.take<uint16_t>(2 * db);
  t.eigh_ws = ar.take<char>(eigh_bytes); t.eigh_ws_bytes = eigh_bytes;
  for (int k = 0; k < 10; ++k) { t.plan_ws[k] = ar.take<char>(plan_bytes[k]); t.plan_ws_bytes[k] = plan_bytes[k]; }
  t.cy_ws = ar.take<char>(1024);
  if (o) *o = t;
  return ar.off;
}

// descriptors of the nine grouped products (8 plans of ps_fd_round_f32 + W W^T / R R^T of the preparation)
void fd_descs(std::vector<ps_gemm_desc> (&ds)[9], const FdCarve& w, int B, int d, int r, int b) {
  auto mk = [&](const float* a, const float* bb_, float* c, int m, int n, int k, int ta, int tb, int64_t lda,
                int64_t ldb, int64_t ldc) {
    ps_gemm_desc g; g.a = a; g.b = bb_; g.c = c; g.m = m; g.n = n; g.k = k; g.transa = ta; g.transb = tb;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; return g;
  };
  for (int k = 0; k < 9; ++k) ds[k].clear();
  for (int j = 0; j < B; ++j) {
    float* x = w.x + (size_t)j * d * b; float* z = w.z + (size_t)j * d * b; float* tmp = w.tmp + (size_t)j * d * b;
    float* g = w.g + (size_t)j * b * b; float* m = w.m + (size_t)j * b * b; float* pol = w.polish + (size_t)j * b * b;
    float* t = w.t + (size_t)j * b * b; float* y = w.y + (size_t)j * b * b;
    float* c = w.c + (size_t)j * d * d;
    ds[0].push_back(mk(x, x, g, b, b, d, 1, 0, b, b, b));        // gram_x: gram = x^T x
    ds[1].push_back(mk(x, m, tmp, d, b, b, 0, 0, b, b, b));      // xm:     tmp = x m
    ds[2].push_back(mk(tmp, tmp, g, b, b, d, 1, 0, b, b, b));    // gram_t: gram = tmp^T tmp
    ds[3].push_back(mk(tmp, pol, x, d, b, b, 0, 0, b, b, b));    // pol:    x = tmp polish
    ds[4].push_back(mk(c, x, z, d, b, d, 0, 0, d, b, b));        // cx:     z = C x   (not used with the x6 planes)
    ds[5].push_back(mk(x, z, t, b, b, d, 1, 0, b, b, b));        // xtz:    t = x^T z
    ds[6].push_back(mk(x, y, tmp, d, b, b, 0, 0, b, b, b));      // xy:     tmp = x y
    ds[7].push_back(mk(z, y, tmp, d, b, b, 0, 0, b, b, b));      // zy:     tmp = z y
    const float* wj = w.weighted + (size_t)j * d * r;
    ds[8].push_back(mk(wj, wj, c, d, d, r, 0, 1, r, r, d));      // C = W W^T
  }
}
}  // namespace
}  // namespace psk

using namespace psk;

extern "C" int ps_fd_block_columns(int rank, int d) { return fd_block_columns(rank, d); }

static int fd_update_supported(const ps_fd_update_desc* u, int* b_out) {
  if (!u || u->batch < 1 || u->d < 1 || u->rank < 1 || u->p < 1) return PS_EINVAL;
  const int B = u->batch, d = u->d, r = u->rank, b = fd_block_columns(r, d);
  if (b_out) *b_out = b;
  // the domain of the fused kernels (ps_fd_cy_step_f32 / ps_fd_cx6_f32 / ps_fd_round_f32)
  if (B > 16 || d < 128 || d % 128 != 0 || !(b == 32 || b == 64 || b == 96) || b > ps_chol_rinv_max_n() ||
      b > ps_eigh_sorted_max_n() || 4 * (r + 33) > d || r + 2 >= d)
    return PS_EUNSUPPORTED;
  return PS_OK;
}

static size_t fd_update_sizes(const ps_fd_update_desc* u, int b, size_t plan_bytes[10], size_t* eigh_bytes) {
  const int B = u->batch, d = u->d, r = u->rank;
  FdCarve dummy{};
  // plan workspaces: sized on descriptors with the real shapes (pointers do not matter for the size)
  std::vector<ps_gemm_desc> ds[9];
  fd_descs(ds, dummy, B, d, r, b);
  for (int k = 0; k < 9; ++k) plan_bytes[k] = ps_gemm_grouped_workspace_bytes(ds[k].data(), B) + 256;
  plan_bytes[9] = 256;
  if (u->input_is_factor) {   // Gram = R R^T: d x d x d per factor
    std::vector<ps_gemm_desc> gd(B);
    for (int j = 0; j < B; ++j) {
      ps_gemm_desc g; memset(&g, 0, sizeof(g));
      g.m = d; g.n = d; g.k = d; g.transa = 0; g.transb = 1; g.lda = d; g.ldb = d; g.ldc = d;
      gd[j] = g;
    }
    plan_bytes[9] = ps_gemm_grouped_workspace_bytes(gd.data(), B) + 256;
  }
  std::vector<int32_t> nn(B, b);
  *eigh_bytes = ps_eigh_root_workspace_bytes(B, nn.data()) + 256;
  psh::Arena ar(nullptr, 0);
  return fd_carve(ar, nullptr, B, d, r, b, u->input_is_factor != 0, plan_bytes, *eigh_bytes) + 256;
}

// ---- one group of factors as a resumable run ---------------------------------------------------------
// begin = preparation + first round; step = the host read of a round, then filter + next round (or done);
// finish = DS:1196-1290 and the packing.  A call with one group is begin, step ... step, finish -- the same commands
// in the same order as the straight-line driver it replaces.  Two groups (B >= 4) run on two streams with their
// steps interleaved: while the host waits for one group's 16-byte read, the other group's round is queued, and the
// latency-bound kernels of a round (chol_rinv / eigh_small: ONE workgroup per factor, 0.7 of the ~3 ms of a round of
// 8 factors) run under the other group's bandwidth-bound filter steps.  Factors are independent: same bits.
namespace {
struct FdRun {
  ps_fd_update_desc u;       // this group's view of the call (pointers advanced to its first factor)
  hipStream_t st = nullptr;
  int b = 0, degree = 0, max_outer = 0;
  float tol = 0.f;
  FdCarve w;
  std::vector<ps_gemm_desc> ds[9];
  ps_gemm_plan* plans[8] = {};
  ps_fd_round_desc rd;
  std::vector<const void*> p0, p1, p2;
  std::vector<ps_gemm_bf16_desc> cy;
  int32_t* h_sum = nullptr;  // pinned, 4 words
  hipEvent_t ev = nullptr;   // behind the summary's copy
  int outer = 0, filter_steps = 0;
  bool done = false;
  ~FdRun() { for (int i = 0; i < 8; ++i) if (plans[i]) (void)ps_gemm_grouped_plan_destroy(plans[i]); }
};

int fd_run_post_round(FdRun& g) {
  PS_HIP(hipMemcpyAsync(g.h_sum, g.w.summary, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, g.st));
  PS_HIP(hipEventRecord(g.ev, g.st));
  return PS_OK;
}

int fd_run_begin(FdRun& g, void* workspace, size_t workspace_bytes) {
  const ps_fd_update_desc* u = &g.u;
  void* stream = (void*)g.st;
  hipStream_t st = g.st;
  const int B = u->batch, d = u->d, r = u->rank, b = g.b, k = r + 1;
  size_t pb[10], eb = 0;
  const size_t need = fd_update_sizes(u, b, pb, &eb);
  if (workspace_bytes < need) return PS_EWORKSPACE;
  psh::Arena ar(workspace, workspace_bytes);
  FdCarve& w = g.w;
  fd_carve(ar, &w, B, d, r, b, u->input_is_factor != 0, pb, eb);
  if (ar.overflow) return PS_EWORKSPACE;
  for (int j = 0; j < B; ++j) if (!u->new_grad[j]) return PS_EINVAL;
  fd_descs(g.ds, w, B, d, r, b);

  // ---- prepare: C = sym(decay W W^T + Gram), bf16 planes ----------------------------------------
  hipLaunchKernelGGL(fd_prep_kernel, dim3(64, B), dim3(256), 0, st, u->prev, w.weighted, d, r, u->ridge_epsilon,
                     u->error_tolerance, u->relative_matrix_epsilon);
  PS_LAUNCH_CHECK();
  PS_RC(ps_gemm_grouped_f32(stream, g.ds[8].data(), B, w.plan_ws[8], w.plan_ws_bytes[8]));
  std::vector<const float*> grams(B);
  if (u->input_is_factor) {   // Gram = R R^T (the reference's statistics slot holds R, DS:1497-1505)
    std::vector<ps_gemm_desc> gd(B);
    for (int j = 0; j < B; ++j) {
      ps_gemm_desc q; q.a = u->new_grad[j]; q.b = u->new_grad[j]; q.c = w.gram + (size_t)j * d * d;
      q.m = d; q.n = d; q.k = d; q.transa = 0; q.transb = 1; q.lda = d; q.ldb = d; q.ldc = d;
      gd[j] = q; grams[j] = q.c;
    }
    PS_RC(ps_gemm_grouped_f32(stream, gd.data(), B, w.plan_ws[9], w.plan_ws_bytes[9]));
  } else {
    for (int j = 0; j < B; ++j) grams[j] = u->new_grad[j];
  }
  PS_RC(ps_fd_cov_update_f32(stream, w.c, grams.data(), B, d, u->decay));
  g.p0.resize(B); g.p1.resize(B); g.p2.resize(B);
  for (int j = 0; j < B; ++j) {
    g.p0[j] = (uint16_t*)w.planes[0] + (size_t)j * d * d;
    g.p1[j] = (uint16_t*)w.planes[1] + (size_t)j * d * d;
    g.p2[j] = (uint16_t*)w.planes[2] + (size_t)j * d * d;
    PS_RC(ps_convert_f32_to_bf16x3_frag(stream, w.c + (size_t)j * d * d, (void*)g.p0[j], (void*)g.p1[j],
                                        (void*)g.p2[j], d, d, d));
  }
  PS_HIP(hipMemcpyAsync(w.x, u->x0, (size_t)B * d * b * sizeof(float), hipMemcpyDeviceToDevice, st));

  // ---- iterate ------------------------------------------------------------------------------------
  for (int q = 0; q < 8; ++q)
    if (q != 4)
      PS_RC(ps_gemm_grouped_plan_create(stream, g.ds[q].data(), B, w.plan_ws[q], w.plan_ws_bytes[q], &g.plans[q]));
  ps_fd_round_desc& rd = g.rd;
  memset(&rd, 0, sizeof(rd));
  rd.gram_x = g.plans[0]; rd.xm = g.plans[1]; rd.gram_t = g.plans[2]; rd.pol = g.plans[3]; rd.cx = nullptr;
  rd.xtz = g.plans[5]; rd.xy = g.plans[6]; rd.zy = g.plans[7];
  rd.c0 = g.p0.data(); rd.c1 = g.p1.data(); rd.c2 = g.p2.data();
  rd.xt0 = w.xt[0]; rd.xt1 = w.xt[1]; rd.xt2 = w.xt[2];
  rd.x = w.x; rd.z = w.z; rd.tmp = w.tmp;
  rd.gram = w.g; rd.m = w.m; rd.polish = w.polish; rd.t = w.t; rd.y = w.y; rd.sym = w.sym;
  rd.evals = w.evals; rd.evecs = w.evecs; rd.theta = w.theta; rd.res = w.res;
  rd.eigh_workspace = w.eigh_ws; rd.eigh_workspace_bytes = w.eigh_ws_bytes;
  rd.params = w.params; rd.converged = w.conv; rd.summary = w.summary;
  rd.batch = B; rd.n = d; rd.b = b; rd.k = k; rd.degree = g.degree; rd.orthonormalize = 1; rd.tol = g.tol;
  g.cy.resize(B);
  for (int j = 0; j < B; ++j) {
    memset(&g.cy[j], 0, sizeof(g.cy[j]));
    g.cy[j].a_hi = g.p0[j]; g.cy[j].a_lo = g.p1[j]; g.cy[j].lda = d; g.cy[j].a_tiled = 2;
    g.cy[j].b_hi = (uint16_t*)w.yt_hi + (size_t)j * d; g.cy[j].b_lo = (uint16_t*)w.yt_lo + (size_t)j * d;
    g.cy[j].c = w.z + (size_t)j * d * b;
    g.cy[j].m = d; g.cy[j].n = b; g.cy[j].k = d; g.cy[j].ldb = (int64_t)B * d; g.cy[j].ldc = b;
  }
  PS_RC(ps_fd_round_f32(stream, &rd));
  g.outer = 1;
  return fd_run_post_round(g);
}

// the one host read of a round, then the next round (or done)
int fd_run_step(FdRun& g) {
  const int B = g.u.batch, d = g.u.d, b = g.b;
  void* stream = (void*)g.st;
  PS_HIP(hipEventSynchronize(g.ev));
  if (g.h_sum[0] || g.outer > g.max_outer) { g.done = true; return PS_OK; }
  const int max_deg = g.h_sum[1];
  int32_t which = -1;
  FdCarve& w = g.w;
  PS_RC(ps_fd_filter_round_f32(stream, g.cy.data(), B, w.z, w.x, w.s0, w.s1, w.yt_hi, w.yt_lo, w.params, max_deg, d, b,
                               (int64_t)B * d, w.cy_ws, 1024, &which));
  g.filter_steps += max_deg > 1 ? max_deg - 1 : 0;
  float* src = which == 0 ? w.x : (which == 1 ? w.s0 : w.s1);
  if (src != w.x)
    PS_HIP(hipMemcpyAsync(w.x, src, (size_t)B * d * b * sizeof(float), hipMemcpyDeviceToDevice, g.st));
  PS_RC(ps_fd_round_f32(stream, &g.rd));
  ++g.outer;
  return fd_run_post_round(g);
}

int fd_run_finish(FdRun& g) {
  const ps_fd_update_desc* u = &g.u;
  const int B = u->batch, d = u->d, r = u->rank, b = g.b;
  FdCarve& w = g.w;
  // (w.tmp, the rounds' [B][d][b] temporary, is free now: the column norms' partial sums go there)
  hipLaunchKernelGGL(fd_colnorm_part_kernel, dim3(FD_NP, B), dim3(256), 0, g.st, w.x, w.tmp, d, b, r);
  const size_t lds = (size_t)((r + 1) + 4 * r) * sizeof(float);
  hipLaunchKernelGGL(fd_finish_kernel, dim3(FD_NS, B), dim3(256), lds, g.st, w.theta, w.x, u->prev, w.tmp, u->out, d, b,
                     r, u->p, u->decay);
  PS_LAUNCH_CHECK();
  PS_HIP(hipMemcpyAsync(u->converged, w.conv, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, g.st));
  return PS_OK;
}

// factors of the first group when a call of B factors runs as two (0: one group)
inline int fd_split(int B) {
  const int want = psh::resolve(nullptr).fd_groups;
  return (want >= 2 && B >= 4) ? (B + 1) / 2 : 0;
}
ps_fd_update_desc fd_sub_desc(const ps_fd_update_desc* u, int j0, int Bg, int b) {
  ps_fd_update_desc s = *u;
  const size_t d = (size_t)u->d, pd = (size_t)u->rank + 2;
  s.batch = Bg;
  s.new_grad = u->new_grad + j0;
  s.prev = u->prev + (size_t)j0 * d * pd;
  s.out = u->out + (size_t)j0 * d * pd;
  s.x0 = u->x0 + (size_t)j0 * d * (size_t)b;
  s.converged = u->converged + j0;
  return s;
}
}  // namespace

extern "C" size_t ps_fd_update_workspace_bytes(const ps_fd_update_desc* u) {
  int b = 0;
  if (fd_update_supported(u, &b) != PS_OK) return 0;
  size_t pb[10], eb = 0;
  const size_t one = fd_update_sizes(u, b, pb, &eb);
  const int B1 = fd_split(u->batch);
  if (B1 == 0) return one;
  const ps_fd_update_desc u1 = fd_sub_desc(u, 0, B1, b), u2 = fd_sub_desc(u, B1, u->batch - B1, b);
  const size_t two = psh::align_up(fd_update_sizes(&u1, b, pb, &eb), 4096) + fd_update_sizes(&u2, b, pb, &eb);
  return two > one ? two : one;
}

extern "C" int ps_fd_update_batched_f32(void* stream, const ps_fd_update_desc* u, int32_t* info_host) {
  PS_DEVICE_CHECK();
  int b = 0;
  int rc = fd_update_supported(u, &b);
  if (rc != PS_OK) return rc;
  if (!u->new_grad || !u->prev || !u->out || !u->converged || !u->x0 || !u->workspace) return PS_EINVAL;
  if (u->workspace_bytes < ps_fd_update_workspace_bytes(u)) return PS_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  static thread_local int32_t* h_sum = nullptr;   // 2 x 4 words, pinned
  static thread_local hipEvent_t evs[4] = {};      // per group: behind the summary copy; fork; join
  if (!h_sum && hipHostMalloc((void**)&h_sum, 8 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) return PS_EINTERNAL;
  for (int i = 0; i < 4; ++i)
    if (!evs[i] && hipEventCreateWithFlags(&evs[i], hipEventDisableTiming) != hipSuccess) return PS_EINTERNAL;
  const int B1 = fd_split(u->batch);
  const int ngroups = B1 > 0 ? 2 : 1;
  hipStream_t side = ngroups == 2 ? psh::side_stream(0) : nullptr;
  if (ngroups == 2 && !side) return PS_EINTERNAL;
  FdRun run[2];
  size_t woff[2] = {0, 0}, wbytes[2] = {u->workspace_bytes, 0};
  if (ngroups == 2) {
    size_t pb[10], eb = 0;
    run[0].u = fd_sub_desc(u, 0, B1, b);
    run[1].u = fd_sub_desc(u, B1, u->batch - B1, b);
    wbytes[0] = psh::align_up(fd_update_sizes(&run[0].u, b, pb, &eb), 4096);
    woff[1] = wbytes[0];
    wbytes[1] = u->workspace_bytes - wbytes[0];
  } else {
    run[0].u = *u;
  }
  for (int q = 0; q < ngroups; ++q) {
    FdRun& g = run[q];
    g.st = q == 0 ? st : side;
    g.b = b;
    g.degree = u->degree > 0 ? u->degree : 12;
    g.max_outer = u->max_outer > 0 ? u->max_outer : 14;
    g.tol = u->tol > 0.f ? u->tol : 1e-5f;
    g.h_sum = h_sum + 4 * q;
    g.ev = evs[q];
  }
  if (ngroups == 2) {   // the side stream starts behind everything queued on the caller's
    PS_HIP(hipEventRecord(evs[2], st));
    PS_HIP(hipStreamWaitEvent(side, evs[2], 0));
  }
  for (int q = 0; q < ngroups && rc == PS_OK; ++q)
    rc = fd_run_begin(run[q], (char*)u->workspace + woff[q], wbytes[q]);
  while (rc == PS_OK && !(run[0].done && (ngroups == 1 || run[1].done))) {
    for (int q = 0; q < ngroups && rc == PS_OK; ++q)
      if (!run[q].done) rc = fd_run_step(run[q]);
  }
  for (int q = 0; q < ngroups && rc == PS_OK; ++q) rc = fd_run_finish(run[q]);
  if (ngroups == 2) {   // the caller's stream continues behind the side stream (also on an error: nothing of this
    (void)hipEventRecord(evs[3], side);   // call may still be running when the workspace is reused)
    (void)hipStreamWaitEvent(st, evs[3], 0);
  }
  if (rc != PS_OK) return rc;
  if (info_host) {
    info_host[0] = run[0].outer; info_host[1] = run[0].filter_steps; info_host[2] = b;
    if (ngroups == 2) {
      if (run[1].outer > info_host[0]) info_host[0] = run[1].outer;
      if (run[1].filter_steps > info_host[1]) info_host[1] = run[1].filter_steps;
    }
  }
  return PS_OK;
}
