// fd.hip — device kernels of the O(d b) / O(b^3) glue of the Frequent-Directions branch
// (BASELINE configs[4]; reference: _fd_update_root DS:1123-1290, whose SVD at DS:1193 this build
// replaces by Chebyshev-filtered subspace iteration, precondition_amd/subspace.py):
//   ps_fd_filter_step_f32   one step of the scaled Chebyshev recurrence for every factor of a
//                           call, fused with the bf16 hi/lo split + transposition of the new
//                           iterate that the next C @ Y product (gemm_bf16.hip) consumes; it
//                           replaces ~8 elementwise torch kernels and a conversion launch per step;
//   ps_chol_rinv_batched_f32  R^-1 of the Cholesky factor G = R^T R of the b x b Gram matrices of
//                           the iterate blocks (float64 in LDS, one workgroup per matrix): the
//                           orthonormalisation X <- X R^-1 (CholeskyQR) that replaces one of the
//                           two small eigendecompositions of every outer round.
// Both are HBM / latency bound byte work; nothing here is reshaped into a GEMM.
#include <math.h>
#include <stdint.h>

#include "common.h"
#include "gemm_core.hip.h"

namespace psk {

// params[j] = {ctr, e, sigma1, deg}.  step 1: y' = (z - ctr y) sigma1 / e.  step k >= 2 with
// sigma = sigma_{k-1} (sigma_1 = sigma1, sigma_m = 1 / (2 / sigma1 - sigma_{m-1})):
// sigma' = 1 / (2 / sigma1 - sigma), y' = (z - ctr y) 2 sigma' / e - sigma sigma' y_prev.
// A factor whose degree is below `step` keeps its iterate (y' = y).
__global__ __launch_bounds__(256) void fd_filter_step_kernel(
    const float* z, const float* y, const float* y_prev, float* y_next, uint16_t* hi,
    uint16_t* lo, const float* params, int step, int n, int b, int64_t ldt, int tiles_r,
    int tiles_c) {
  __shared__ float t[64][65];
  const int per = tiles_r * tiles_c;
  const int j = blockIdx.x / per, rem = blockIdx.x % per;
  const int tr = rem / tiles_c, tc = rem % tiles_c;
  const int r0 = tr * 64, c0 = tc * 64, tid = threadIdx.x;
  const float ctr = params[4 * j + 0], e = params[4 * j + 1], sigma1 = params[4 * j + 2];
  const int deg = (int)params[4 * j + 3];
  float c1, c2 = 0.f;
  if (step <= 1) {
    c1 = sigma1 / e;
  } else {
    float sigma = sigma1;
    for (int m = 2; m < step; ++m) sigma = 1.f / (2.f / sigma1 - sigma);
    const float sn = 1.f / (2.f / sigma1 - sigma);
    c1 = 2.f * sn / e;
    c2 = sigma * sn;
  }
  const bool active = step <= deg;
  const int64_t base = (int64_t)j * n * b;
  for (int el = tid; el < 64 * 64; el += 256) {
    const int r = el >> 6, c = el & 63;
    float v = 0.f;
    if (r0 + r < n && c0 + c < b) {
      const int64_t o = base + (int64_t)(r0 + r) * b + c0 + c;
      const float yy = gload1(y + o);
      v = yy;
      if (active) {
        v = (gload1(z + o) - ctr * yy) * c1;
        if (step > 1) v -= c2 * gload1(y_prev + o);
      }
      gstore1(y_next + o, v);
    }
    t[r][c] = v;
  }
  if (hi == nullptr) return;
  __syncthreads();
  if (ldt == 0) {
    // fragment-major planes of the fused step (gemm_bf16.hip fd_cy_step_kernel; n % 64 == 0,
    // b % 32 == 0): per factor [n / 16][b / 32][lane = 32 (k / 8 % 2) + col % 32][8 k]
    const int cbs = b >> 5;
    for (int it = tid; it < 4 * 2 * 64; it += 256) {
      const int ln = it & 63, cbl = (it >> 6) & 1, kk = it >> 7;
      const int c = c0 + cbl * 32 + (ln & 31);
      if (c >= b) continue;
      const int rb = kk * 16 + (ln >> 5) * 8;
      const int64_t o = base + ((int64_t)((r0 >> 4) + kk) * cbs + (c >> 5)) * 512 + ln * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float x = t[rb + i][c - c0];
        const __bf16 h = (__bf16)x;
        hi[o + i] = __builtin_bit_cast(uint16_t, h);
        if (lo != nullptr) lo[o + i] = __builtin_bit_cast(uint16_t, (__bf16)(x - (float)h));
      }
    }
    return;
  }
  for (int el = tid; el < 64 * 64; el += 256) {
    const int c = el >> 6, r = el & 63;    // output row = column c of the iterate
    if (r0 + r >= n || c0 + c >= b) continue;
    const float x = t[r][c];
    const __bf16 h = (__bf16)x;
    const int64_t o = (int64_t)(c0 + c) * ldt + (int64_t)j * n + r0 + r;
    hi[o] = __builtin_bit_cast(uint16_t, h);
    if (lo != nullptr) lo[o] = __builtin_bit_cast(uint16_t, (__bf16)(x - (float)h));
  }
}

// One workgroup per Gram matrix G (b x b, b <= 96).  L = chol(G) (lower, float64); a pivot at
// or below drop_rel * max diag(G) drops its direction (row and column of the result zero: the
// column of X it belongs to becomes zero, like the eigen-based orthonormalisation drops the
// directions of a rank-deficient block).  out = R^-1 = (L^-1)^T, float32, row-major [b][b].
//
// Both halves are right-looking rank-1 sweeps over LDS with ONE barrier per column and no
// dependent FMA chain (thread = row r x column parity h; the value every lane multiplies by is one
// broadcast LDS read).  The left-looking / column-per-thread form before it spent its time in
// chains of float64 FMAs that each waited for two LDS reads: 0.22 ms for 8 x 96^2.
//   factor:  for c: p = S[c][c];  L[r][c] = S[r][c] / sqrt(p) (kept in a second array, so that the
//            unscaled column stays readable);  S[r][j] -= S[r][c] S[j][c] / p   (c < j <= r)
//   invert:  X = I;  for k: X[k][:] /= L[k][k] (folded into the multiplier);
//            X[r][c] -= (L[r][k] / L[k][k]) X[k][c]   (r > k, c <= k)
constexpr int CQ_MAX = 96;
__global__ __launch_bounds__(256) void chol_rinv_kernel(const float* gram, float* out, int b,
                                                        float drop_rel) {
  extern __shared__ double cq[];
  double* S = cq;                       // [b][b + 1]: the trailing matrix, then X = L^-1
  double* Lm = cq + CQ_MAX * (CQ_MAX + 1);   // [b][b + 1]: L
  __shared__ double s_max;
  __shared__ int s_drop[CQ_MAX];
  const int tid = threadIdx.x, ldS = b + 1;
  const int r = tid & 127, h = tid >> 7;
  const float* g = gram + (int64_t)blockIdx.x * b * b;
  for (int e = tid; e < b * b; e += 256) {
    const int rr = e / b, c = e % b;
    S[rr * ldS + c] = 0.5 * ((double)g[rr * b + c] + (double)g[c * b + rr]);
  }
  __syncthreads();
  if (tid == 0) {
    double m = 0.0;
    for (int i = 0; i < b; ++i) m = fmax(m, S[i * ldS + i]);
    s_max = m;
  }
  __syncthreads();
  const double thresh = (double)drop_rel * s_max;
  for (int c = 0; c < b; ++c) {
    const double p = S[c * ldS + c];
    const bool ok = p > thresh && p > 0.0 && p < 1.0e300;
    // 1 / sqrt(p) from the hardware seed + two Newton steps (a float64 division or square root is a
    // ~50-instruction sequence, and every wavefront would run three of them per column)
    double rs = __builtin_amdgcn_rsq(ok ? p : 1.0);
    rs = rs * fma(-0.5 * (ok ? p : 1.0) * rs, rs, 1.5);
    rs = rs * fma(-0.5 * (ok ? p : 1.0) * rs, rs, 1.5);
    if (r >= c && r < b) {
      double* sr = S + r * ldS;
      const double src = sr[c];
      if (h == 0) {
        Lm[r * ldS + c] = r == c ? (ok ? p * rs : 1.0) : (ok ? src * rs : 0.0);
        if (r == c) s_drop[c] = ok ? 0 : 1;
      }
      if (ok && r > c) {
        const double m = -(src * rs) * rs;         // - S[r][c] / p
        const double* sc = S + c;
        int j = c + 1 + h;
        // four independent element updates per trip: the loads of a trip are in flight together
        for (; j + 6 <= r; j += 8) {
          const double a0 = sr[j], a1 = sr[j + 2], a2 = sr[j + 4], a3 = sr[j + 6];
          const double b0 = sc[j * ldS], b1 = sc[(j + 2) * ldS], b2 = sc[(j + 4) * ldS],
                       b3 = sc[(j + 6) * ldS];
          sr[j] = fma(m, b0, a0);
          sr[j + 2] = fma(m, b1, a1);
          sr[j + 4] = fma(m, b2, a2);
          sr[j + 6] = fma(m, b3, a3);
        }
        for (; j <= r; j += 2) sr[j] = fma(m, sc[j * ldS], sr[j]);
      }
    }
    __syncthreads();
  }
  // X = L^-1 in S (lower); dropped directions: zero row and column
  for (int e = tid; e < b * b; e += 256) {
    const int rr = e / b, c = e % b;
    S[rr * ldS + c] = (rr == c && !s_drop[c]) ? 1.0 : 0.0;
  }
  __syncthreads();
  // 1 / L[k][k] once per row (the diagonal of Lm becomes its reciprocal)
  if (tid < b) Lm[tid * ldS + tid] = 1.0 / Lm[tid * ldS + tid];
  __syncthreads();
  for (int k = 0; k < b; ++k) {
    // row k of X is final up to its 1 / L[k][k]; rows below subtract their multiple of it
    if (r > k && r < b && !s_drop[r] && !s_drop[k]) {
      const double m = -Lm[r * ldS + k] * Lm[k * ldS + k];
      double* sr = S + r * ldS;
      const double* sk = S + k * ldS;
      int c = h;
      for (; c + 6 <= k; c += 8) {
        const double a0 = sr[c], a1 = sr[c + 2], a2 = sr[c + 4], a3 = sr[c + 6];
        const double b0 = sk[c], b1 = sk[c + 2], b2 = sk[c + 4], b3 = sk[c + 6];
        sr[c] = fma(m, b0, a0);
        sr[c + 2] = fma(m, b1, a1);
        sr[c + 4] = fma(m, b2, a2);
        sr[c + 6] = fma(m, b3, a3);
      }
      for (; c <= k; c += 2) sr[c] = fma(m, sk[c], sr[c]);
    }
    __syncthreads();
  }
  float* o = out + (int64_t)blockIdx.x * b * b;
  for (int e = tid; e < b * b; e += 256) {
    const int rr = e / b, c = e % b;      // out[rr][c] = R^-1[rr][c] = L^-1[c][rr] / L[c][c]
    o[e] = (float)(S[c * ldS + rr] * Lm[c * ldS + c]);
  }
}


// C_j <- 0.5 ((decay C_j + G_j) + (decay C_j + G_j)^T) in place, for the stacked covariances of a
// group of factors (DS:1174-1193's operand: decay * W W^T + R R^T, symmetrised): one pass over C and
// G (12 bytes per element) instead of an axpy per factor, a strided transposed add and a scaling
// pass over the whole stack (~5 x the traffic).  One workgroup owns the tile pair (ti, tj), (tj, ti).
struct CovArgs { const float* g[16]; float* c; int n, tiles, pairs; float decay; };
__global__ __launch_bounds__(256) void fd_cov_update_kernel(const CovArgs a) {
  __shared__ float sa[64][65], sb[64][65];
  const int j = blockIdx.x / a.pairs;
  int rem = blockIdx.x % a.pairs, ti = 0;
  while (rem >= a.tiles - ti) { rem -= a.tiles - ti; ++ti; }   // pair index -> (ti, tj >= ti)
  const int tj = ti + rem, tid = threadIdx.x, n = a.n;
  float* c = a.c + (int64_t)j * n * n;
  const float* g = a.g[j];
  const int r0 = ti * 64, c0 = tj * 64;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, q = e & 63;
    float u = 0.f, v = 0.f;
    if (r0 + r < n && c0 + q < n) {
      const int64_t o = (int64_t)(r0 + r) * n + c0 + q;
      const float t = a.decay * gload1(c + o);
      u = t + gload1(g + o);
    }
    if (ti != tj && c0 + r < n && r0 + q < n) {
      const int64_t o = (int64_t)(c0 + r) * n + r0 + q;
      const float t = a.decay * gload1(c + o);
      v = t + gload1(g + o);
    }
    sa[r][q] = u;
    sb[r][q] = v;
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, q = e & 63;
    if (ti == tj) {
      if (r0 + r < n && c0 + q < n) gstore1(c + (int64_t)(r0 + r) * n + c0 + q, 0.5f * (sa[r][q] + sa[q][r]));
    } else {
      if (r0 + r < n && c0 + q < n) gstore1(c + (int64_t)(r0 + r) * n + c0 + q, 0.5f * (sa[r][q] + sb[q][r]));
      if (c0 + r < n && r0 + q < n) gstore1(c + (int64_t)(c0 + r) * n + r0 + q, 0.5f * (sa[q][r] + sb[r][q]));
    }
  }
}

// ---- small kernels of ps_fd_round_f32 (the b x b / n x b glue of one outer round) ----------------
// one thread per element of the [batch][b][b] arrays (grid = batch * ceil(b * b / 256))
// which 0: polish = 1.5 I - 0.5 gram;  1: sym = (t + t^T) / 2;
// 2: eigenpairs ascending (eigh_small_kernel) -> descending: theta[i] = evals[b-1-i], y[:, i] = evecs[:, b-1-i]
__global__ __launch_bounds__(256) void fd_bb_kernel(const float* in, const float* in2, float* out,
                                                    float* out2, int b, int which, int per) {
  const int j = blockIdx.x / per, e = (blockIdx.x % per) * 256 + threadIdx.x;
  if (e >= b * b) return;
  const int64_t base = (int64_t)j * b * b;
  const int r = e / b, c = e % b;
  if (which == 0) {
    const float h = -0.5f * in[base + e];
    out[base + e] = (r == c ? 1.5f : 0.f) + h;
  } else if (which == 1) {
    out[base + e] = 0.5f * (in[base + e] + in[base + (int64_t)c * b + r]);
  } else {
    out[base + e] = in[base + (int64_t)r * b + (b - 1 - c)];
    if (r == 0) out2[(int64_t)j * b + c] = in2[(int64_t)j * b + (b - 1 - c)];
  }
}
// res[j][c] = || z[j][:, c] - theta[j][c] x[j][:, c] ||_2 in two launches: partial sums of squares over
// FD_RCH row chunks (one workgroup per factor, 32 columns and chunk), then the chunks in order
constexpr int FD_RCH = 32;
__global__ __launch_bounds__(256) void fd_resnorm_part_kernel(const float* x, const float* z,
                                                              const float* theta, float* part, int n,
                                                              int b) {
  __shared__ float sh[8][33];
  const int cbs = (b + 31) / 32;
  const int ch = blockIdx.x % FD_RCH, cb = (blockIdx.x / FD_RCH) % cbs, j = blockIdx.x / (FD_RCH * cbs);
  const int c = cb * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  const int rows = (n + FD_RCH - 1) / FD_RCH, r0 = ch * rows, r1 = min(n, r0 + rows);
  const int64_t base = (int64_t)j * n * b;
  float acc = 0.f;
  if (c < b) {
    const float th = theta[(int64_t)j * b + c];
    for (int r = r0 + rg; r < r1; r += 8) {
      const float dlt = gload1(z + base + (int64_t)r * b + c) - gload1(x + base + (int64_t)r * b + c) * th;
      acc += dlt * dlt;
    }
  }
  sh[rg][threadIdx.x & 31] = acc;
  __syncthreads();
  if (rg == 0 && c < b) {
    float s0 = 0.f;
    for (int g = 0; g < 8; ++g) s0 += sh[g][threadIdx.x & 31];
    part[((int64_t)j * FD_RCH + ch) * b + c] = s0;
  }
}
__global__ __launch_bounds__(256) void fd_resnorm_sum_kernel(const float* part, float* res, int b, int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int j = e / b, c = e % b;
  float s0 = 0.f;
  for (int ch = 0; ch < FD_RCH; ++ch) s0 += part[((int64_t)j * FD_RCH + ch) * b + c];
  res[e] = sqrtf(s0);
}

// Per-round control of the subspace iteration (precondition_amd/subspace.py), one thread per
// factor: convergence of the k wanted Ritz pairs, the scaled Chebyshev filter's interval and
// per-factor degree (theta_1 may be amplified over theta_k by at most ~1e2 = exp(4.6)), written
// as the params rows ps_fd_filter_step_f32 reads.  summary = {all converged, max degree, min
// degree, largest wanted relative residual above 2e-2}: ONE small host read per round instead of
// ~40 elementwise launches and two.
__device__ inline float fd_acosh(float v) {
  v = fminf(fmaxf(v, 1.f), 1e30f);
  return logf(v) + log1pf(sqrtf(fmaxf(1.f - 1.f / (v * v), 0.f)));
}
__global__ void fd_round_control_kernel(const float* theta, const float* res, int batch, int b,
                                        int k, int n, float tol, int degree, float* params,
                                        int* converged, int* summary) {
  __shared__ int s_all, s_max, s_min, s_plain;
  if (threadIdx.x == 0) { s_all = 1; s_max = 1; s_min = degree; s_plain = 0; }
  __syncthreads();
  for (int j = threadIdx.x; j < batch; j += blockDim.x) {
    const float* th = theta + (int64_t)j * b;
    const float* rs = res + (int64_t)j * b;
    const float top = fmaxf(th[0], 1e-30f);
    bool conv = true, plain = false;
    for (int i = 0; i < k; ++i) {
      const bool wanted = th[i] > n * 2.4e-7f * top;
      conv = conv && (rs[i] <= tol * top || !wanted);
      plain = plain || (rs[i] / top > 2e-2f);
    }
    const float cut = fmaxf(th[b - 1], 0.f);
    const float e = fmaxf(0.5f * cut, 1e-30f * top), ctr = 0.5f * cut;
    const float a0 = top * (1.0f + 1e-6f);
    const float sigma1 = e / (a0 - ctr);
    const float xk = fmaxf(th[k - 1], 1e-30f * top);
    const float spread = fmaxf(fd_acosh((top - ctr) / e) - fd_acosh((xk - ctr) / e), 1e-6f);
    const float deg = fminf(fmaxf(floorf(4.6f / spread), 1.f), (float)degree);
    params[4 * j + 0] = ctr; params[4 * j + 1] = e; params[4 * j + 2] = sigma1; params[4 * j + 3] = deg;
    converged[j] = conv ? 1 : 0;
    if (!conv) atomicAnd(&s_all, 0);
    atomicMax(&s_max, (int)deg);
    atomicMin(&s_min, (int)deg);
    if (plain) atomicOr(&s_plain, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) { summary[0] = s_all; summary[1] = s_max; summary[2] = s_min; summary[3] = s_plain; }
}

}  // namespace psk

using namespace psk;

extern "C" int ps_fd_filter_step_f32(void* stream, const float* z, const float* y,
                                     const float* y_prev, float* y_next, void* yt_hi,
                                     void* yt_lo, const float* params, int step, int batch,
                                     int64_t n, int64_t b, int64_t ldt) {
  PS_DEVICE_CHECK();
  if (!z || !y || !y_next || !params || batch < 1 || n < 1 || b < 1 || step < 1 ||
      (step > 1 && !y_prev) || (yt_hi && ldt != 0 && ldt < (int64_t)batch * n) || (!yt_hi && yt_lo))
    return PS_EINVAL;
  if (yt_hi && ldt == 0 && (n % 64 != 0 || b % 32 != 0)) return PS_EUNSUPPORTED;
  const int64_t tr = (n + 63) / 64, tc = (b + 63) / 64;
  if (tr * tc * batch > 0x7fffffff) return PS_EUNSUPPORTED;
  hipLaunchKernelGGL(fd_filter_step_kernel, dim3((unsigned)(tr * tc * batch)), dim3(256), 0,
                     (hipStream_t)stream, z, y, y_prev, y_next, (uint16_t*)yt_hi,
                     (uint16_t*)yt_lo, params, step, (int)n, (int)b, ldt, (int)tr, (int)tc);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_fd_cov_update_f32(void* stream, float* c, const float* const* gram, int batch,
                                    int64_t n, float decay) {
  PS_DEVICE_CHECK();
  if (!c || !gram || batch < 1 || n < 1) return PS_EINVAL;
  const int64_t tiles = (n + 63) / 64, pairs = tiles * (tiles + 1) / 2;
  if (tiles > 32767 || pairs * 16 > 0x7fffffff) return PS_EUNSUPPORTED;
  for (int j0 = 0; j0 < batch; j0 += 16) {
    CovArgs a{};
    const int nb = batch - j0 < 16 ? batch - j0 : 16;
    for (int j = 0; j < nb; ++j) {
      if (!gram[j0 + j]) return PS_EINVAL;
      a.g[j] = gram[j0 + j];
    }
    a.c = c + (int64_t)j0 * n * n;
    a.n = (int)n; a.tiles = (int)tiles; a.pairs = (int)pairs; a.decay = decay;
    hipLaunchKernelGGL(fd_cov_update_kernel, dim3((unsigned)(pairs * nb)), dim3(256), 0,
                       (hipStream_t)stream, a);
    PS_LAUNCH_CHECK();
  }
  return PS_OK;
}

extern "C" int ps_fd_round_f32(void* stream, const ps_fd_round_desc* d) {
  PS_DEVICE_CHECK();
  if (!d || d->batch < 1 || d->n < 1 || d->b < 1 || d->k < 1 || d->k > d->b || !d->x || !d->z || !d->tmp ||
      !d->gram || !d->m || !d->polish || !d->t || !d->y || !d->sym || !d->evals || !d->evecs || !d->theta ||
      !d->res || !d->eigh_workspace || !d->params || !d->converged || !d->summary || !d->xtz || !d->xy ||
      !d->zy || (d->orthonormalize && (!d->gram_x || !d->xm || !d->gram_t || !d->pol)))
    return PS_EINVAL;
  const bool x6 = d->c0 != nullptr;
  if (x6 ? (!d->c1 || !d->c2 || !d->xt0 || !d->xt1 || !d->xt2) : !d->cx) return PS_EINVAL;
  // the eigenpairs must come back sorted (the LDS-resident solver of b <= 128 sorts them)
  if (d->b > CQ_MAX || d->b > ps_eigh_sorted_max_n()) return PS_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int B = d->batch, b = d->b, per = (b * b + 255) / 256;
  if (b < FD_RCH) return PS_EUNSUPPORTED;   // the residual partials are kept in `sym`
  const size_t blk_bytes = (size_t)B * d->n * b * sizeof(float);
  if (d->orthonormalize) {
    PS_RC(ps_gemm_grouped_plan_launch(stream, d->gram_x));
    PS_RC(ps_chol_rinv_batched_f32(stream, d->gram, d->m, b, B, 1e-10f));
    PS_RC(ps_gemm_grouped_plan_launch(stream, d->xm));
    PS_RC(ps_gemm_grouped_plan_launch(stream, d->gram_t));
    hipLaunchKernelGGL(fd_bb_kernel, dim3(B * per), dim3(256), 0, st, d->gram, nullptr, d->polish, nullptr, b, 0, per);
    PS_RC(ps_gemm_grouped_plan_launch(stream, d->pol));
  }
  if (x6) PS_RC(ps_fd_cx6_f32(stream, d->c0, d->c1, d->c2, B, d->x, d->z, d->xt0, d->xt1, d->xt2, d->n, b));
  else PS_RC(ps_gemm_grouped_plan_launch(stream, d->cx));
  PS_RC(ps_gemm_grouped_plan_launch(stream, d->xtz));
  hipLaunchKernelGGL(fd_bb_kernel, dim3(B * per), dim3(256), 0, st, d->t, nullptr, d->sym, nullptr, b, 1, per);
  {
    std::vector<const float*> a(B);
    std::vector<float*> ev(B), vv(B);
    std::vector<int32_t> nn(B, b);
    for (int j = 0; j < B; ++j) {
      a[j] = d->sym + (size_t)j * b * b;
      ev[j] = d->evals + (size_t)j * b;
      vv[j] = d->evecs + (size_t)j * b * b;
    }
    PS_RC(ps_eigh_batched_f32(stream, a.data(), nn.data(), nn.data(), B, ev.data(), vv.data(), nn.data(),
                              d->eigh_workspace, d->eigh_workspace_bytes));
  }
  hipLaunchKernelGGL(fd_bb_kernel, dim3(B * per), dim3(256), 0, st, d->evecs, d->evals, d->y, d->theta, b, 2, per);
  PS_RC(ps_gemm_grouped_plan_launch(stream, d->xy));
  PS_HIP(hipMemcpyAsync(d->x, d->tmp, blk_bytes, hipMemcpyDeviceToDevice, st));
  PS_RC(ps_gemm_grouped_plan_launch(stream, d->zy));
  PS_HIP(hipMemcpyAsync(d->z, d->tmp, blk_bytes, hipMemcpyDeviceToDevice, st));
  // the partial sums live in `sym` (free after the eigensolver): batch * 32 * b <= batch * b * b floats
  hipLaunchKernelGGL(fd_resnorm_part_kernel, dim3(B * ((b + 31) / 32) * FD_RCH), dim3(256), 0, st, d->x, d->z,
                     d->theta, d->sym, d->n, b);
  hipLaunchKernelGGL(fd_resnorm_sum_kernel, dim3((B * b + 255) / 256), dim3(256), 0, st, d->sym, d->res, b, B * b);
  PS_LAUNCH_CHECK();
  return ps_fd_round_control_f32(stream, d->theta, d->res, B, b, d->k, d->n, d->tol, d->degree, d->params,
                                 d->converged, d->summary);
}

extern "C" int ps_chol_rinv_max_n(void) { return CQ_MAX; }

extern "C" int ps_chol_rinv_batched_f32(void* stream, const float* gram, float* out, int b,
                                        int batch, float drop_rel) {
  PS_DEVICE_CHECK();
  if (!gram || !out || batch < 1 || b < 1 || drop_rel < 0.f) return PS_EINVAL;
  if (b > CQ_MAX) return PS_EUNSUPPORTED;
  const size_t lds = 2 * (size_t)CQ_MAX * (CQ_MAX + 1) * sizeof(double);
  static bool attr = false;
  if (!attr) {
    PS_HIP(hipFuncSetAttribute((const void*)chol_rinv_kernel,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  hipLaunchKernelGGL(chol_rinv_kernel, dim3((unsigned)batch), dim3(256), lds,
                     (hipStream_t)stream, gram, out, b, drop_rel);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_fd_round_control_f32(void* stream, const float* theta, const float* res,
                                       int batch, int b, int k, int n, float tol, int degree,
                                       float* params, int32_t* converged, int32_t* summary) {
  PS_DEVICE_CHECK();
  if (!theta || !res || !params || !converged || !summary || batch < 1 || b < 1 || k < 1 || k > b ||
      degree < 1)
    return PS_EINVAL;
  hipLaunchKernelGGL(fd_round_control_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, theta, res,
                     batch, b, k, n, tol, degree, params, converged, summary);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
