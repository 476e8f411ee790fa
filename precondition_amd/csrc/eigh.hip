// eigh.hip — batched inverse p-th root by symmetric eigendecomposition
// (reference: matrix_inverse_pth_root_eigh, DS:943-1030; jnp.linalg.eigh at DS:1007
// is LAPACK ssyevd on the reference's CPU path).
//
// Three eigensolvers share this driver:
//   n <= 128                     eigh_small_kernel: LDS-resident one-sided Jacobi, one launch;
//   root mode, n > 128 (round 3) eigh_cj.hip.h: one-sided block Jacobi on the float64-accumulated
//                                Cholesky factor (no eigenvector accumulation, 6 n^3 per sweep);
//   its fallback (Cholesky breakdown) and plain eigenpairs of possibly indefinite matrices
//   (ps_eigh_batched_f32, n > 128): the blocked two-sided Jacobi described next.
//
// Blocked two-sided Jacobi, built for MFMA + LDS instead of
// translating a tridiagonalisation:
//   * the matrix is cut into 64-wide block columns; a round pairs them up
//     (round-robin tournament, nb/2 disjoint pairs, nb-1 rounds per sweep);
//   * jacobi_pair_kernel: one workgroup per (matrix, pair) pulls the 128x128
//     pivot submatrix into LDS and runs one cyclic sweep of scalar Jacobi
//     rotations on it there (127 rounds x 64 disjoint rotations, parallel
//     ordering), accumulating the 128x128 orthogonal factor Q in LDS;
//   * jacobi_row_kernel / jacobi_col_kernel apply A <- J^T A, A <- A J, V <- V J
//     (J = blockdiag of the Q's) as 128x128-tile fp32-MFMA products with K = 128
//     gathered from the two block rows / columns (gemm_core, in place).
//   Sweeps run until the pivot off-norm drops below 1e-3 of ||A||_F; V is then
//   re-orthogonalised by one Newton-Schulz step (V <- V (1.5 I - 0.5 V^T V)), A is
//   recomputed as V^T D V, and one or two more sweeps finish quadratically.  This
//   keeps V orthogonal to ~1e-6 (rotations applied hundreds of times in float32
//   otherwise drift to ~1e-4) and gives roots within LAPACK's own float32 error
//   of the float64 answer (prototype: tools/proto_block_jacobi.py).
// The root itself follows DS:1005-1021: eps = ridge*max(max_ev, tol); inv_e = e==0 ? 0
// : max(e, eps)^(-1/p); val = (u sqrt(inv_e))(u sqrt(inv_e))^T; error = max|u^T D u - diag(e)|.
// Eigenvector order/sign is not unique, so parity is checked on val, never on u.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "options.h"
#include "gemm_core.hip.h"
#include "gemm_bf16x.hip.h"
#include "power_iter.hip.h"

namespace psk {

constexpr int EBK = 16;
constexpr int JB = 64;          // block-column width
constexpr int JP = 2 * JB;      // pivot size = TILE
constexpr int JLD = JP + 1;     // LDS row stride (odd: column walks are conflict-free)

enum EPhase { EP_SWEEP = 0, EP_CONVERGED = 1 };

struct EighBlock {
  const float* a;
  float* out;
  float* A;   // working matrix (becomes ~diagonal)
  float* V;   // accumulated eigenvectors
  float* D;   // regularised input (kept for the polish and the error metric)
  float* W;   // temp
  float* X;   // temp
  float* Q;   // [npairs][128*128] rotation factors of the current round
  float* offpart;  // [npairs * rounds] pivot off-norm^2 partials of the current sweep
  float* sumsq_partial;
  float* evals;
  float* evals_out;  // mode 1 only
  int n, n_full, lda, ldo, npad, nb, npairs, p;
  float alpha, ridge, max_ev, normD;
  int active, sweeps;
  int small;   // n <= 128: solved by the LDS-resident one-sided Jacobi kernel, never swept
  float off_rel;
  unsigned err_bits;
  int power_iters;
  unsigned soff_bits;  // max scaled off-diagonal |a_ij| / sqrt(|a_ii a_jj|) (eigh_scaled_off_kernel)
  // one-sided block Jacobi on the Cholesky factor (eigh_cj.hip.h)
  int cj;          // this block takes that path
  int cj_active;   // ... and is still sweeping
  int chol_fail;   // the factorisation met a non-positive pivot: two-sided fallback
  int cj_group;    // which of the two interleaved streams sweeps this block
  int td_done;     // eigh_td.hip.h solved this block (its result stands; the Jacobi solvers skip it)
};

struct ETile {
  int block;
  short k, t;   // pair index within the round, tile index along the long dimension
  short which;  // col phase: 0 = A, 1 = V
  short pad_;
};

struct EStatus {
  int gen;
  int active;
  float max_off;
  int pad_;
};

// Round-robin tournament: pair k of round r among m (even) players.
__device__ __host__ inline void rr_pair(int m, int r, int k, int& I, int& J) {
  int a, b;
  if (k == 0) { a = m - 1; b = r % (m - 1); }
  else { a = (r + k) % (m - 1); b = (r - k + (m - 1)) % (m - 1); }
  I = a < b ? a : b;
  J = a < b ? b : a;
}

// Same tournament without ordering the two players (the in-LDS pivot sweep walks
// the first players (r+k) % (m-1) and the second players (r-k) % (m-1) with lanes).
// 0 <= r < m-1 and 0 <= k < m/2, so the reductions are one conditional subtraction each
// (an integer division by a run-time value costs ~40 instructions per lane and round).
__device__ inline void rr_pair_raw(int m, int r, int k, int& a, int& b) {
  const int mm = m - 1;
  if (k == 0) { a = mm; b = r; }
  else {
    a = r + k; if (a >= mm) a -= mm;
    b = r - k; if (b < 0) b += mm;
  }
}

// ---- init: D = A_in masked + ridge I, A = D, V = I; ||D||_F partials -------------
__global__ __launch_bounds__(256) void eigh_init_kernel(EighBlock* blocks,
                                                        const ETile* tiles, int skip_td_done) {
  __shared__ float red[4];
  const ETile te = tiles[blockIdx.x];
  EighBlock* eb = &blocks[te.block];
  // second start of the blocks eigh_td.hip.h handed back: its own results and the blocks the LDS-resident
  // solver has already finished stay as they are
  if (skip_td_done && (eb->td_done || eb->small)) return;
  const int n = eb->n, ld = eb->npad, tid = threadIdx.x;
  const int tpr = ld / TILE;
  const float ridge = eb->ridge;
  float ss = 0.f;
  for (int e = tid; e < TILE * TILE; e += 256) {
    const int row = te.k * TILE + e / TILE, col = te.t * TILE + e % TILE;
    float d = 0.f;
    if (row < n && col < n) {
      d = eb->a[(int64_t)row * eb->lda + col];
      if (row == col) d = __fadd_rn(d, ridge);  // DS:1006
      ss += d * d;
    }
    const int64_t o = (int64_t)row * ld + col;
    eb->D[o] = d;
    eb->A[o] = d;
    eb->V[o] = row == col ? 1.f : 0.f;
  }
  ss = wave_sum_f32(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  if (tid == 0) eb->sumsq_partial[te.k * tpr + te.t] = ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ void eigh_setup_kernel(EighBlock* blocks, const PiBlock* pis, int nblocks,
                                  float ridge_epsilon, float tol, int relative) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  EighBlock* eb = &blocks[b];
  float max_ev = 1.f;
  int pit = 0;
  if (relative) { max_ev = pis[b].lambda; pit = pis[b].iters; }
  eb->max_ev = max_ev;
  eb->power_iters = pit;
  eb->ridge = __fmul_rn(ridge_epsilon, fmaxf(max_ev, tol));  // DS:1005
  if (max_ev != max_ev) eb->ridge = max_ev;
}

// ---- one cyclic Jacobi sweep on the 128x128 pivot, in LDS --------------------------
constexpr int JT = 1024;  // threads of the pivot kernel: 16 wavefronts hide LDS latency
__global__ __launch_bounds__(JT) void jacobi_pair_kernel(EighBlock* blocks,
                                                         const ETile* tiles, int round,
                                                         int sweep_slot) {
  extern __shared__ __align__(16) float jsm[];
  float* S = jsm;                 // [128][129]
  float* Q = jsm + JP * JLD;      // [128][129]
  float* cs = Q + JP * JLD;       // [64][2]
  __shared__ float red[JT / 64];
  const ETile te = tiles[blockIdx.x];
  EighBlock* eb = &blocks[te.block];
  if (!eb->active || round >= eb->nb - 1) return;
  int I, J;
  rr_pair(eb->nb, round, te.k, I, J);
  const int ld = eb->npad, tid = threadIdx.x;
  const float* A = eb->A;
  for (int e = tid; e < JP * JP; e += JT) {
    const int r = e >> 7, c = e & 127;
    const int gr = r < JB ? I * JB + r : J * JB + (r - JB);
    const int gc = c < JB ? I * JB + c : J * JB + (c - JB);
    S[r * JLD + c] = gload1(A + (int64_t)gr * ld + gc);
    Q[r * JLD + c] = r == c ? 1.f : 0.f;
  }
  __syncthreads();
  // symmetrise (A is symmetric only to rounding) and measure the pivot off-norm
  float off = 0.f;
  for (int e = tid; e < JP * JP; e += JT) {
    const int r = e >> 7, c = e & 127;
    if (r < c) {
      const float v = 0.5f * (S[r * JLD + c] + S[c * JLD + r]);
      S[r * JLD + c] = v;
      S[c * JLD + r] = v;
      // the whole strict upper triangle: the diagonal blocks' own off-diagonals
      // count too (they are the only ones there are when n <= 64); they are
      // re-counted in every round of the sweep, which only makes the test stricter
      off += 2.f * v * v;
    }
  }
  off = wave_sum_f32(off);
  if ((tid & 63) == 0) red[tid >> 6] = off;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < JT / 64; ++w) t += red[w];
    eb->offpart[round * eb->npairs + te.k] = t;
  }

  // One round = 64 disjoint rotations (p_k, q_k).  Every 2x2 block {p_k,q_k} x {p_k',q_k'}
  // of S belongs to exactly one thread, which applies rotation k from the left and k'
  // from the right in registers (no barrier between the two sides).  Lanes run over
  // k': the round-robin columns (r+k') % 127 and (r-k') % 127 are consecutive in a
  // row, so the LDS walks are bank-conflict free.
  float2* cs2 = reinterpret_cast<float2*>(cs);
  // wave index as a scalar: the row pairs of this wave are then SALU work
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = JT / 64;
  for (int rr = 0; rr < JP - 1; ++rr) {
    if (tid < JB) {
      int p, q;
      rr_pair_raw(JP, rr, tid, p, q);
      const float app = S[p * JLD + p], aqq = S[q * JLD + q], apq = S[p * JLD + q];
      float c = 1.f, s = 0.f;
      if (fabsf(apq) > 1e-30f) {
        const float tau = (aqq - app) / (2.f * apq);
        const float t = copysignf(1.f, tau) / (fabsf(tau) + sqrtf(1.f + tau * tau));
        c = 1.f / sqrtf(1.f + t * t);
        s = t * c;
        if (!(c == c) || !(s == s)) { c = 1.f; s = 0.f; }
      }
      cs2[tid] = make_float2(c, s);
    }
    __syncthreads();
    int ca, cb;
    rr_pair_raw(JP, rr, lane, ca, cb);
    const float2 cc = cs2[lane];
    // all loads first (the compiler cannot reorder LDS loads over possibly aliasing
    // stores itself), then the arithmetic, then all stores
    constexpr int NB = JB / NW, NQ = JP / NW;
    float x00[NB], x01[NB], x10[NB], x11[NB], qx[NQ], qy[NQ];
    float2 cr[NB];
    int ra[NB], rb[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int k = wave + NW * i;
      rr_pair_raw(JP, rr, k, ra[i], rb[i]);
      cr[i] = cs2[k];
      const float* s0 = S + ra[i] * JLD;
      const float* s1 = S + rb[i] * JLD;
      x00[i] = s0[ca]; x01[i] = s0[cb]; x10[i] = s1[ca]; x11[i] = s1[cb];
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const float* q0 = Q + (wave + NW * i) * JLD;
      qx[i] = q0[ca]; qy[i] = q0[cb];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const float y00 = cr[i].x * x00[i] - cr[i].y * x10[i], y10 = cr[i].y * x00[i] + cr[i].x * x10[i];
      const float y01 = cr[i].x * x01[i] - cr[i].y * x11[i], y11 = cr[i].y * x01[i] + cr[i].x * x11[i];
      float* s0 = S + ra[i] * JLD;
      float* s1 = S + rb[i] * JLD;
      s0[ca] = cc.x * y00 - cc.y * y01;
      s0[cb] = cc.y * y00 + cc.x * y01;
      s1[ca] = cc.x * y10 - cc.y * y11;
      s1[cb] = cc.y * y10 + cc.x * y11;
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      float* q0 = Q + (wave + NW * i) * JLD;
      q0[ca] = cc.x * qx[i] - cc.y * qy[i];
      q0[cb] = cc.y * qx[i] + cc.x * qy[i];
    }
    __syncthreads();
  }
  float* Qg = eb->Q + (int64_t)te.k * JP * JP;
  for (int e = tid; e < JP * JP; e += JT) {
    const int r = e >> 7, c = e & 127;
    gstore1(Qg + e, Q[r * JLD + c]);
  }
}

// ---- LDS-resident eigensolver for n <= 128: the whole decomposition in ONE launch ------
// One 1024-thread workgroup per matrix; G = A V (starts as A) and V (starts as I) live in
// LDS column-major (column stride 132 floats: 16-byte aligned, 2 x 66 KB), and all sweeps
// and the convergence test run inside the kernel (no host round trip, any batch size).
// One-sided (Hestenes) Jacobi: a rotation of the column pair (p, q) of G and V makes g_p
// and g_q orthogonal; when all pairs are, the columns of V are the eigenvectors of the
// symmetric input and lambda_j = v_j . g_j (its Rayleigh quotient, sign included).  Unlike the
// two-sided form a rotation touches only its own two columns, so a round of m/2 disjoint
// pairs (round-robin tournament) needs a single barrier, every LDS access is a 16-byte read
// or write of a contiguous column piece, and the three dot products of a pair are reduced
// inside a quarter-wavefront: 16 lanes own one pair (8 elements of each column per lane), a
// wavefront 4 pairs, the 16 wavefronts the 64 pairs of a round at n = 128.
// LDS traffic per round = G and V read and written once (2 x 64 KB at n = 128), the bound of
// the kernel (ds_write_b128 ~79 B/clk): ~2.2k cycles per round, ~8 sweeps of n-1 rounds.
// Replaces, for small matrices, the multi-launch blocked driver below (jnp.linalg.eigh at
// DS:1007 / DS:1071 and the b x b problems of subspace.py).
constexpr int SE_T = 1024;          // threads
constexpr int SE_LD = 132;          // column stride (floats)
constexpr int SE_MAXN = 128;
constexpr int SE_MAX_SWEEPS = 24;

// x + (x rotated right by N lanes inside its row of 16 lanes): one v_add_f32 with a DPP
// operand.  Rotations by 8, 4, 2, 1 leave the sum of the 16 lanes in every lane.
template <int N>
__device__ inline float row16_add_ror(float x) {
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + N, 0xf, 0xf, false);
  return x + __int_as_float(r);
}
__device__ inline float row16_sum(float x) {
  x = row16_add_ror<8>(x);
  x = row16_add_ror<4>(x);
  x = row16_add_ror<2>(x);
  return row16_add_ror<1>(x);
}

// All sweeps of the one-sided Jacobi on the LDS images G (= A V) and V, m (even) players,
// 1024 threads.  Returns the number of sweeps run (the last one without a rotation).
// EXTRA = elements per lane beyond the first 64 rows: 0 (m <= 64), 2 (m <= 96: rows 64 + 2l,
// 64 + 2l + 1), 4 (m <= 128: rows 64 + 4l ...).  Rows that do not exist are never touched: a
// 96 x 96 problem (the block size of the subspace iteration) does 3/4 of the 128-row work.
// done_cos2: the iteration also stops after a sweep whose largest rotated pair had a squared
// cosine below it (the NEXT sweep would be the quadratically converged, rotation-free one): 0
// keeps the strict rule.  Used by the pivots of eigh_cj.hip.h, whose outer iteration re-checks
// every pair against the true Gram matrix anyway.
template <int EXTRA>
__device__ inline int onesided_jacobi_lds_t(float* G, float* V, int* s_rot, int m,
                                            int max_sweeps, float done_cos2 = 0.f) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane >> 4, l = lane & 15;     // 16 lanes per pair, 4 pairs per wavefront
  const int k = 4 * wave + sub;                 // pair index within the round
  const bool has_pair = k < (m >> 1);
  int sweeps_total = 0;
  for (int sweeps = 0; sweeps < max_sweeps; ++sweeps, ++sweeps_total) {
    float rotated = 0.f;   // largest squared cosine of a rotated pair in this sweep
    for (int round = 0; round < m - 1; ++round) {
      if (has_pair) {
        int p, q;
        rr_pair_raw(m, round, k, p, q);
        float* gp = G + p * SE_LD + 4 * l;
        float* gq = G + q * SE_LD + 4 * l;
        float* vp = V + p * SE_LD + 4 * l;
        float* vq = V + q * SE_LD + 4 * l;
        // elements 4l..4l+3 and 64+4l..64+4l+3 of each column (16 lanes x 16 B contiguous);
        // all eight reads are issued before anything waits on them
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 a0 = *reinterpret_cast<f32x4*>(gp), b0 = *reinterpret_cast<f32x4*>(gq);
        f32x4 w0 = *reinterpret_cast<f32x4*>(vp), x0 = *reinterpret_cast<f32x4*>(vq);
        f32x4 a1 = zero4, b1 = zero4, w1 = zero4, x1 = zero4;
        if (EXTRA == 4) {
          a1 = *reinterpret_cast<f32x4*>(gp + 64); b1 = *reinterpret_cast<f32x4*>(gq + 64);
          w1 = *reinterpret_cast<f32x4*>(vp + 64); x1 = *reinterpret_cast<f32x4*>(vq + 64);
        } else if (EXTRA == 2) {
          // rows 64 + 2l, 64 + 2l + 1: column base + 64 + 2l = (base + 4l) + 64 - 2l
          const f32x2 ta = *reinterpret_cast<f32x2*>(gp + 64 - 2 * l), tb = *reinterpret_cast<f32x2*>(gq + 64 - 2 * l);
          const f32x2 tw = *reinterpret_cast<f32x2*>(vp + 64 - 2 * l), tx = *reinterpret_cast<f32x2*>(vq + 64 - 2 * l);
          a1[0] = ta[0]; a1[1] = ta[1]; b1[0] = tb[0]; b1[1] = tb[1];
          w1[0] = tw[0]; w1[1] = tw[1]; x1[0] = tx[0]; x1[1] = tx[1];
        }
        float aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          aa += a0[j] * a0[j]; bb += b0[j] * b0[j]; ab += a0[j] * b0[j];
        }
#pragma unroll
        for (int j = 0; j < EXTRA; ++j) {
          aa += a1[j] * a1[j]; bb += b1[j] * b1[j]; ab += a1[j] * b1[j];
        }
        aa = row16_sum(aa); bb = row16_sum(bb); ab = row16_sum(ab);
        // rotate unless the columns are already orthogonal to working precision
        if (fabsf(ab) > 3e-7f * __builtin_amdgcn_sqrtf(aa * bb)) {
          const float zeta = (bb - aa) * __builtin_amdgcn_rcpf(2.f * ab);
          const float t = copysignf(1.f, zeta) *
                          __builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(1.f + zeta * zeta));
          const float c = __builtin_amdgcn_rsqf(1.f + t * t), sn = c * t;
          if (c == c && sn == sn) {
            f32x4 na0, nb0, nw0, nx0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              na0[j] = c * a0[j] - sn * b0[j]; nb0[j] = sn * a0[j] + c * b0[j];
              nw0[j] = c * w0[j] - sn * x0[j]; nx0[j] = sn * w0[j] + c * x0[j];
            }
            *reinterpret_cast<f32x4*>(gp) = na0; *reinterpret_cast<f32x4*>(gq) = nb0;
            *reinterpret_cast<f32x4*>(vp) = nw0; *reinterpret_cast<f32x4*>(vq) = nx0;
            if (EXTRA > 0) {
              f32x4 na1 = zero4, nb1 = zero4, nw1 = zero4, nx1 = zero4;
#pragma unroll
              for (int j = 0; j < EXTRA; ++j) {
                na1[j] = c * a1[j] - sn * b1[j]; nb1[j] = sn * a1[j] + c * b1[j];
                nw1[j] = c * w1[j] - sn * x1[j]; nx1[j] = sn * w1[j] + c * x1[j];
              }
              if (EXTRA == 4) {
                *reinterpret_cast<f32x4*>(gp + 64) = na1; *reinterpret_cast<f32x4*>(gq + 64) = nb1;
                *reinterpret_cast<f32x4*>(vp + 64) = nw1; *reinterpret_cast<f32x4*>(vq + 64) = nx1;
              } else {
                *reinterpret_cast<f32x2*>(gp + 64 - 2 * l) = f32x2{na1[0], na1[1]};
                *reinterpret_cast<f32x2*>(gq + 64 - 2 * l) = f32x2{nb1[0], nb1[1]};
                *reinterpret_cast<f32x2*>(vp + 64 - 2 * l) = f32x2{nw1[0], nw1[1]};
                *reinterpret_cast<f32x2*>(vq + 64 - 2 * l) = f32x2{nx1[0], nx1[1]};
              }
            }
            rotated = fmaxf(rotated, ab * ab * __builtin_amdgcn_rcpf(aa * bb));
          }
        }
      }
      __syncthreads();
    }
    // a sweep without a single rotation: converged (s_rot is double buffered by parity)
    if (rotated > 0.f && l == 0) atomicMax(&s_rot[sweeps & 1], __float_as_int(rotated));
    __syncthreads();
    const int any = s_rot[sweeps & 1];
    __syncthreads();
    if (tid == 0) s_rot[sweeps & 1] = 0;   // re-armed for sweep + 2 (after the next barriers)
    if (!any || __int_as_float(any) < done_cos2) { ++sweeps_total; break; }
  }
  __syncthreads();
  return sweeps_total;
}

__device__ inline int onesided_jacobi_lds(float* G, float* V, int* s_rot, int m, int extra,
                                          int max_sweeps) {
  if (extra == 0) return onesided_jacobi_lds_t<0>(G, V, s_rot, m, max_sweeps);
  if (extra == 2) return onesided_jacobi_lds_t<2>(G, V, s_rot, m, max_sweeps);
  return onesided_jacobi_lds_t<4>(G, V, s_rot, m, max_sweeps);
}

// Renormalises the columns of V (the approximate rcp / rsq of the rotation parameters scale a
// rotation by 1 + O(eps)) and of G with them (G = A V).  1024 threads; ends with a barrier.
__device__ inline void onesided_renormalize(float* G, float* V, int m) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = wave; j < m; j += SE_T / 64) {
    const float v0 = V[j * SE_LD + lane], v1 = V[j * SE_LD + 64 + lane];
    const float nn = wave_sum_f32(v0 * v0 + v1 * v1);
    const float inv = nn > 0.f ? 1.f / sqrtf(nn) : 0.f;
    V[j * SE_LD + lane] = v0 * inv; V[j * SE_LD + 64 + lane] = v1 * inv;
    G[j * SE_LD + lane] *= inv; G[j * SE_LD + 64 + lane] *= inv;
  }
  __syncthreads();
}

__global__ __launch_bounds__(SE_T) void eigh_small_kernel(EighBlock* blocks, const int* ids,
                                                          int refresh) {
  extern __shared__ __align__(16) float sem[];
  float* G = sem;                       // [128][132] column-major: G[col * SE_LD + row]
  float* V = sem + SE_MAXN * SE_LD;
  float* s_red = sem + 2 * SE_MAXN * SE_LD;                 // [16] per-wavefront partials
  int* s_rot = reinterpret_cast<int*>(s_red + 16);          // [2]
  EighBlock* eb = &blocks[ids[blockIdx.x]];
  const int n = eb->n, ld = eb->npad, tid = threadIdx.x;
  if (n == 0) return;
  const int m = (n + 1) & ~1;           // players of the tournament (a dummy column if n is odd)
  // elements per lane beyond row 63: rows 64.. do not exist / exist up to 95 / up to 127
  const int extra = n <= 64 ? 0 : (n <= 96 ? 2 : 4);
  const float* A = eb->A;               // regularised input D (eigh_init_kernel), stride npad
  const int lane = tid & 63, wave = tid >> 6;
  float shift = 0.f;
  int sweeps_total = 0;
  // One-sided Jacobi yields the SVD: for an indefinite matrix with eigenvalues +x and -x of
  // close magnitude the singular subspace mixes their eigenvectors.  Such inputs (detected
  // from the Rayleigh quotients) are solved again as A + shift I, which is positive definite.
  for (int attempt = 0; attempt < 2; ++attempt) {
    // G = (D + D^T)/2 + shift I column-major, zero padded to 128 rows / m columns; V = I
    for (int e = tid; e < SE_MAXN * SE_MAXN; e += SE_T) {
      const int col = e >> 7, row = e & 127;
      float g = 0.f;
      if (row < n && col < n) {
        g = 0.5f * (gload1(A + (int64_t)col * ld + row) + gload1(A + (int64_t)row * ld + col));
        if (row == col) g += shift;
      }
      G[col * SE_LD + row] = g;
      V[col * SE_LD + row] = row == col ? 1.f : 0.f;
    }
    if (tid < 2) s_rot[tid] = 0;
    __syncthreads();

    sweeps_total += onesided_jacobi_lds(G, V, s_rot, m, extra, SE_MAX_SWEEPS);
    // Refresh: the columns g_j = A v_j were carried through every rotation in float32, i.e.
    // with absolute errors of eps32 * ||A|| -- far above a g_j that belongs to an eigenvalue
    // 1e-6 ||A|| of a graded spectrum, so the rotations among the small directions were
    // computed from noise.  G = A V is recomputed from the converged V with float64
    // accumulation (A staged in G's own LDS image) and the sweeps run on: the dot products
    // of accurate g_j resolve the small directions (root error on graded 64..128 problems
    // 5e-4 ... 2e-3 -> LAPACK-float32 level).  Converged well-conditioned problems pay one
    // rotation-free sweep.  Only spectra graded over more than three decades need it (the
    // carried error is eps32 * ||A||): min ||g_j|| < 1e-3 max ||g_j|| over the real columns.
    bool graded = false;
    if (refresh) {
      float lo2 = 3.0e38f, hi2 = 0.f;
      for (int j = wave; j < n; j += SE_T / 64) {
        const float g0 = G[j * SE_LD + lane], g1 = G[j * SE_LD + 64 + lane];
        const float nn = wave_sum_f32(g0 * g0 + g1 * g1);
        lo2 = fminf(lo2, nn); hi2 = fmaxf(hi2, nn);
      }
      if (lane == 0) { s_red[wave] = lo2; }
      __syncthreads();
      float lo_all = 3.0e38f;
      for (int w = 0; w < SE_T / 64; ++w) lo_all = fminf(lo_all, s_red[w]);
      __syncthreads();
      if (lane == 0) s_red[wave] = hi2;
      __syncthreads();
      float hi_all = 0.f;
      for (int w = 0; w < SE_T / 64; ++w) hi_all = fmaxf(hi_all, s_red[w]);
      __syncthreads();
      graded = lo_all < 1e-6f * hi_all;   // squared norms
    }
    if (graded) {
      for (int e = tid; e < SE_MAXN * SE_MAXN; e += SE_T) {
        const int col = e >> 7, row = e & 127;
        float g = 0.f;
        if (row < n && col < n) {
          g = 0.5f * (gload1(A + (int64_t)col * ld + row) + gload1(A + (int64_t)row * ld + col));
          if (row == col) g += shift;
        }
        G[col * SE_LD + row] = g;    // the (symmetric) matrix itself, column-major
      }
      __syncthreads();
      double ng[SE_MAXN * SE_MAXN / SE_T];
#pragma unroll
      for (int i = 0; i < SE_MAXN * SE_MAXN / SE_T; ++i) {
        const int e = tid + SE_T * i;
        const int col = e >> 7, row = e & 127;
        double acc = 0.0;
        if (col < m && row < n)
          for (int k = 0; k < n; ++k)   // A[row][k] = A[k][row] (symmetrised above)
            acc = fma((double)G[k * SE_LD + row], (double)V[col * SE_LD + k], acc);
        ng[i] = acc;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < SE_MAXN * SE_MAXN / SE_T; ++i) {
        const int e = tid + SE_T * i;
        G[(e >> 7) * SE_LD + (e & 127)] = (float)ng[i];
      }
      if (tid < 2) s_rot[tid] = 0;
      __syncthreads();
      sweeps_total += onesided_jacobi_lds(G, V, s_rot, m, extra, SE_MAX_SWEEPS);
    }
    // The approximate rcp / rsq of the rotation parameters scale a rotation by 1 + O(eps):
    // renormalise the eigenvectors (and g_j with them, G = A V), then the Rayleigh quotients.
    float lo_ev = 0.f, hi_abs = 0.f;
    for (int j = wave; j < m; j += SE_T / 64) {
      const float v0 = V[j * SE_LD + lane], v1 = V[j * SE_LD + 64 + lane];
      const float nn = wave_sum_f32(v0 * v0 + v1 * v1);
      const float inv = nn > 0.f ? 1.f / sqrtf(nn) : 0.f;
      const float g0 = G[j * SE_LD + lane] * inv, g1 = G[j * SE_LD + 64 + lane] * inv;
      V[j * SE_LD + lane] = v0 * inv; V[j * SE_LD + 64 + lane] = v1 * inv;
      G[j * SE_LD + lane] = g0; G[j * SE_LD + 64 + lane] = g1;
      const float ev = wave_sum_f32(v0 * inv * g0 + v1 * inv * g1) - shift;
      // ||g_j|| = the singular value |lambda_j + shift|: exact even when eigenvectors of +x and
      // -x are mixed, so 1.01 max_j ||g_j|| is a safe shift
      const float gn = sqrtf(wave_sum_f32(g0 * g0 + g1 * g1));
      if (j < n) { lo_ev = fminf(lo_ev, ev); hi_abs = fmaxf(hi_abs, gn); }
    }
    if (lane == 0) { s_red[wave] = lo_ev; }
    __syncthreads();
    float lo_all = 0.f;
    for (int w = 0; w < SE_T / 64; ++w) lo_all = fminf(lo_all, s_red[w]);
    __syncthreads();
    if (lane == 0) s_red[wave] = hi_abs;
    __syncthreads();
    float hi_all = 0.f;
    for (int w = 0; w < SE_T / 64; ++w) hi_all = fmaxf(hi_all, s_red[w]);
    __syncthreads();
    // positive semi-definite (the statistics, Gram and Rayleigh-Ritz matrices of this
    // library): done.  Clearly indefinite: once more on A + 1.01 max|lambda| I.
    if (attempt == 1 || !(lo_all < -1e-5f * hi_all)) break;
    shift = 1.01f * hi_all;
  }
  if (eb->evals_out != nullptr) {
    // plain eigenpairs (ps_eigh_batched_f32): straight to the caller's arrays in LAPACK's
    // ASCENDING order.  rank_j = #{i : lambda_i < lambda_j or (equal and i < j)}; the
    // eigenvalue of column j sits in s_ev[j], its rank in s_rank[j] (behind V in the LDS).
    float* s_ev = s_red + 32;                           // [128]
    int* s_rank = reinterpret_cast<int*>(s_ev + 128);   // [128]
    for (int j = wave; j < n; j += SE_T / 64) {
      float d = V[j * SE_LD + lane] * G[j * SE_LD + lane] +
                V[j * SE_LD + 64 + lane] * G[j * SE_LD + 64 + lane];
      d = wave_sum_f32(d);
      if (lane == 0) s_ev[j] = d - shift;
    }
    __syncthreads();
    if (tid < n) {
      const float mine = s_ev[tid];
      int rank = 0;
      for (int i = 0; i < n; ++i) {
        const float o = s_ev[i];
        rank += (o < mine || (o == mine && i < tid)) ? 1 : 0;
      }
      if (mine != mine) {   // NaNs last, in index order
        rank = 0;
        for (int i = 0; i < n; ++i) rank += (s_ev[i] == s_ev[i] || i < tid) ? 1 : 0;
      }
      s_rank[tid] = rank;
      gstore1(eb->evals_out + rank, mine);
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += SE_T) {
      const int col = e / n, row = e - col * n;   // LDS walks a column; the store scatters rows
      gstore1(eb->out + (int64_t)row * eb->ldo + s_rank[col], V[col * SE_LD + row]);
    }
    if (tid == 0) { eb->sweeps = sweeps_total; eb->active = 0; eb->off_rel = 0.f; }
    return;
  }
  // results to the block's workspace: V row-major (eigenvectors in columns), A = diag(lambda)
  float* Aout = eb->A;
  float* Vout = eb->V;
  for (int e = tid; e < SE_MAXN * SE_MAXN; e += SE_T) {
    const int row = e >> 7, col = e & 127;   // consecutive threads: consecutive columns of a row
    const bool in = row < n && col < n;
    gstore1(Vout + (int64_t)row * ld + col, in ? V[col * SE_LD + row] : (row == col ? 1.f : 0.f));
    if (row != col) gstore1(Aout + (int64_t)row * ld + col, 0.f);
  }
  for (int j = wave; j < n; j += SE_T / 64) {
    float d = V[j * SE_LD + lane] * G[j * SE_LD + lane] +
              V[j * SE_LD + 64 + lane] * G[j * SE_LD + 64 + lane];
    d = wave_sum_f32(d);
    if (lane == 0) gstore1(Aout + (int64_t)j * ld + j, d - shift);
  }
  if (tid == 0) { eb->sweeps = sweeps_total; eb->active = 0; eb->off_rel = 0.f; }
}

// ---- A <- J^T A on block rows {I, J}; tile = 128 gathered rows x 128 columns -----
__global__ __launch_bounds__(256, 2) void jacobi_row_kernel(EighBlock* blocks,
                                                            const ETile* tiles, int ntiles,
                                                            int round) {
  __shared__ __align__(16) float smem[SmemCfg<EBK>::TOTAL];
  const ETile te = tiles[xcd_remap(blockIdx.x, ntiles)];
  EighBlock* eb = &blocks[te.block];
  if (!eb->active || round >= eb->nb - 1) return;
  int I, J;
  rr_pair(eb->nb, round, te.k, I, J);
  const int ld = eb->npad;
  const float* Qg = eb->Q + (int64_t)te.k * JP * JP;
  float* A = eb->A;
  f32x16 acc[2][2];
  zero_acc(acc);
  for (int seg = 0; seg < 2; ++seg) {
    const int rb = (seg == 0 ? I : J) * JB;
    Operand a{Qg + seg * JB * JP, JP, 0, JP, JB, true};               // Q^T: (m,k) = Q[k][m]
    Operand b{A + (int64_t)rb * ld, ld, te.t * TILE, ld, JB, true};    // (col,k) = A[rb+k][col]
    gemm_tile_accum<MC, MC, EBK, false>(a, b, JB, smem, acc);
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = acc_row(wm, i, r, lane);
        const int row = m < JB ? I * JB + m : J * JB + (m - JB);
        const int col = te.t * TILE + acc_col(wn, j, lane);
        gstore1(A + (int64_t)row * ld + col, acc[i][j][r]);
      }
}

// ---- X <- X J on block columns {I, J}, X in {A, V} ---------------------------------
__global__ __launch_bounds__(256, 2) void jacobi_col_kernel(EighBlock* blocks,
                                                            const ETile* tiles, int ntiles,
                                                            int round) {
  __shared__ __align__(16) float smem[SmemCfg<EBK>::TOTAL];
  const ETile te = tiles[xcd_remap(blockIdx.x, ntiles)];
  EighBlock* eb = &blocks[te.block];
  if (!eb->active || round >= eb->nb - 1) return;
  int I, J;
  rr_pair(eb->nb, round, te.k, I, J);
  const int ld = eb->npad;
  const float* Qg = eb->Q + (int64_t)te.k * JP * JP;
  float* X = te.which ? eb->V : eb->A;
  f32x16 acc[2][2];
  zero_acc(acc);
  for (int seg = 0; seg < 2; ++seg) {
    const int cb = (seg == 0 ? I : J) * JB;
    Operand a{X + cb, ld, te.t * TILE, ld, JB, true};         // (m,k) = X[rt+m][cb+k]
    Operand b{Qg + seg * JB * JP, JP, 0, JP, JB, true};       // (j,k) = Q[seg*64+k][j]
    gemm_tile_accum<KC, MC, EBK, false>(a, b, JB, smem, acc);
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = te.t * TILE + acc_row(wm, i, r, lane);
        const int c = acc_col(wn, j, lane);
        const int col = c < JB ? I * JB + c : J * JB + (c - JB);
        gstore1(X + (int64_t)row * ld + col, acc[i][j][r]);
      }
}

// ---- per-sweep control ------------------------------------------------------------
// mode 0: after init (norm of D).  mode 1: after a sweep: off_rel of the pivots as
// they were at the START of that sweep; blocks under `tol` stop sweeping.
__global__ __launch_bounds__(256) void eigh_control_kernel(EighBlock* blocks, int nblocks,
                                                           int mode, float tol, int gen,
                                                           EStatus* status) {
  __shared__ int s_act;
  __shared__ float s_max;
  if (threadIdx.x == 0) { s_act = 0; s_max = 0.f; }
  __syncthreads();
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    EighBlock* eb = &blocks[b];
    if (eb->n == 0) continue;
    if (mode == 0) {
      const int t = eb->npad / TILE;
      float ss = 0.f;
      for (int i = 0; i < t * t; ++i) ss += eb->sumsq_partial[i];
      eb->normD = sqrtf(ss);
    } else if (mode == 2) {
      // after eigh_scaled_off_kernel: blocks whose largest scaled off-diagonal entry is still
      // above `tol` sweep once more (NaN: stop, everything downstream is NaN anyway)
      const float so = __uint_as_float(eb->soff_bits);
      eb->soff_bits = 0;
      eb->off_rel = so;
      eb->active = (!eb->small && !eb->td_done && so > tol) ? 1 : 0;
    } else if (eb->active) {
      const int cnt = eb->npairs * (eb->nb - 1);
      float off = 0.f;
      for (int i = 0; i < cnt; ++i) off += eb->offpart[i];
      eb->off_rel = sqrtf(off) / eb->normD;
      eb->sweeps += 1;
      // NaN input: stop (everything downstream is NaN, as in the reference)
      if (!(eb->off_rel >= tol)) eb->active = 0;
    }
    if (eb->active) {
      atomicAdd(&s_act, 1);
      atomicMax(reinterpret_cast<int*>(&s_max), __float_as_int(fmaxf(eb->off_rel, 0.f)));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && status) {
    status->active = s_act;
    status->max_off = s_max;
    __threadfence_system();
    status->gen = gen;
  }
}

__global__ void eigh_set_active_kernel(EighBlock* blocks, int nblocks, int swap_vw) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  EighBlock* eb = &blocks[b];
  if (eb->n == 0) return;
  eb->active = (eb->small || eb->cj || eb->td_done) ? 0 : 1;
  if (swap_vw) { float* t = eb->V; eb->V = eb->W; eb->W = t; }
}

// ---- generic grouped product for the polish / finalisation ---------------------------
enum GEpi { GE_STORE = 0, GE_POLISH = 1, GE_ERR = 2, GE_SYM_STORE = 3, GE_ERR_UPPER = 4 };
enum GBuf { GB_A = 0, GB_V, GB_D, GB_W, GB_X, GB_OUT };

__device__ inline float* ebuf(EighBlock* eb, int id) {
  switch (id) {
    case GB_A: return eb->A;
    case GB_V: return eb->V;
    case GB_D: return eb->D;
    case GB_W: return eb->W;
    case GB_X: return eb->X;
    default: return eb->out;
  }
}

template <int LA, int LB>
__global__ __launch_bounds__(256, 2) void eigh_gemm_kernel(EighBlock* blocks,
                                                           const ETile* tiles, int ntiles,
                                                           int a_id, int b_id, int c_id,
                                                           int epi) {
  __shared__ __align__(16) float smem[SmemCfg<EBK>::TOTAL];
  const ETile te = tiles[xcd_remap(blockIdx.x, ntiles)];
  EighBlock* eb = &blocks[te.block];
  const int ld = eb->npad;
  // symmetric results (W W^T; V^T (D V) up to rounding): the upper tile triangle only.  GE_SYM_STORE
  // mirrors a tile into its transposed position (element (j, i) of W W^T sums the same products in the
  // same order as (i, j): the mirrored value is bit-identical to a computed one).
  if ((epi == GE_SYM_STORE || epi == GE_ERR_UPPER) && te.k > te.t) return;
  Operand A{ebuf(eb, a_id), ld, te.k * TILE, ld, ld, true};
  Operand B{ebuf(eb, b_id), ld, te.t * TILE, ld, ld, true};
  f32x16 acc[2][2];
  // The error metric's own product (DS:1017: u^T (D u)) is summed in segments of 128 (gemm_core.hip.h SEG_K):
  // a diagonal entry is lambda_j as a sum of n positive terms, and ONE float32 chain over k = 2048 rounds
  // it by ~eps sqrt(n) lambda / 2 -- 0.02-0.03 at lambda_max 1.2e4, which IS the metric of a cfg3 block
  // (the eigenvectors are orthogonal to 1.3e-6, a true ssyevd's to 2.3e-6: tools/dev_r6_orth.py), against
  // 0.011-0.015 for the reference's K-blocked sgemm.  Blocked summation brings it below that.  The other
  // products of this kernel keep the single chain (same bits as before).
  const bool seg = epi == GE_ERR || epi == GE_ERR_UPPER;
  gemm_tile<LA, LB, EBK, false, false, false, true>(A, B, ld, smem, acc, nullptr, seg);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  float* C = ebuf(eb, c_id);
  unsigned emax = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = te.k * TILE + acc_row(wm, i, r, lane);
        const int col = te.t * TILE + acc_col(wn, j, lane);
        const float v = acc[i][j][r];
        if (epi == GE_STORE || epi == GE_SYM_STORE) {
          if (c_id == GB_OUT) {
            if (row < eb->n_full && col < eb->n_full)
              gstore1(C + (int64_t)row * eb->ldo + col, v);
          } else {
            gstore1(C + (int64_t)row * ld + col, v);
          }
        } else if (epi == GE_POLISH) {
          gstore1(C + (int64_t)row * ld + col, (row == col ? 1.5f : 0.f) - 0.5f * v);
        } else if (epi == GE_ERR || epi == GE_ERR_UPPER) {
          // DS:1017-1021: |u^T D u - diag(e)|, restricted to the unpadded part
          if (row < eb->n && col < eb->n) {
            const float d = row == col ? eb->evals[row] : 0.f;
            const unsigned e = abs_bits(__fsub_rn(v, d));
            emax = e > emax ? e : emax;
          }
        }
      }
  if (epi == GE_SYM_STORE && te.k != te.t) {
    // mirrored tile through LDS (two 64-row halves, stride 129: conflict-free both ways), guarded
    // stores in 256-byte runs
    constexpr int TLD = 129;
    const int n_out = c_id == GB_OUT ? eb->n_full : ld;
    const int ldc = c_id == GB_OUT ? eb->ldo : ld;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __syncthreads();
      if (wm == h) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int lr = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              smem[lr * TLD + acc_col(wn, j, lane)] = acc[i][j][r];
            }
      }
      __syncthreads();
      for (int e = tid; e < 128 * 64; e += NTHREADS) {
        const int c = e >> 6, lr = e & 63;
        const int orow = te.t * TILE + c, ocol = te.k * TILE + h * 64 + lr;
        if (orow < n_out && ocol < n_out) gstore1(C + (int64_t)orow * ldc + ocol, smem[lr * TLD + c]);
      }
    }
  }
  if (epi == GE_ERR || epi == GE_ERR_UPPER) {
    __syncthreads();
    emax = wave_max_u32(emax);
    unsigned* red = reinterpret_cast<unsigned*>(smem);
    if (lane == 0) red[wave] = emax;
    __syncthreads();
    if (tid == 0) {
      unsigned m = red[0];
      m = red[1] > m ? red[1] : m;
      m = red[2] > m ? red[2] : m;
      m = red[3] > m ? red[3] : m;
      atomicMax(&eb->err_bits, m);
    }
  }
}

// e = diag(A); W = V * sqrt(inv_e) column-wise (DS:1012-1015).
__global__ __launch_bounds__(256) void eigh_scale_kernel(EighBlock* blocks,
                                                         const ETile* tiles) {
  const ETile te = tiles[blockIdx.x];
  EighBlock* eb = &blocks[te.block];
  const int ld = eb->npad, tid = threadIdx.x;
  const float ridge = eb->ridge, alpha = eb->alpha;
  for (int e = tid; e < TILE * TILE; e += 256) {
    const int row = te.k * TILE + e / TILE, col = te.t * TILE + e % TILE;
    const float ev = eb->A[(int64_t)col * ld + col];
    float inv = 0.f;
    if (!(ev == 0.f)) inv = powf(fmaxf(ev, ridge), alpha);
    if (ev != ev || ridge != ridge) inv = __uint_as_float(0x7fc00000u);
    eb->W[(int64_t)row * ld + col] = eb->V[(int64_t)row * ld + col] * sqrtf(inv);
    if (row == col) eb->evals[col] = ev;
  }
}

// ---- eigenvalue refinement: e_i = v_i^T D v_i accumulated in float64 -------------------
// The Jacobi diagonal carries the solver's absolute error (a few eps * ||D||), which is a
// several-percent relative error on eigenvalues at the ridge (1e-6 * lambda_max) — exactly
// where max(e, eps)^(-1/p) is steep.  The Rayleigh quotient of the converged vectors with
// float64 accumulation is accurate to the vectors' second-order error.  One workgroup per
// (matrix, 64 eigenvectors): X = D V_I in 64-row panels, 4x4 float64 accumulators per
// thread, operands staged through LDS as float32; n^3 DFMA per matrix on the vector pipe
// (3 % of the time at 64 x 2048^2).  The refined values replace diag(A).
constexpr int RQ = 64;   // panel: 64 rows x 64 eigenvectors
constexpr int RK = 16;   // k chunk
__global__ __launch_bounds__(256) void eigh_rayleigh_f64_kernel(EighBlock* blocks,
                                                                 const ETile* tiles) {
  __shared__ float sD[RQ][RK + 1];
  __shared__ float sV[RK][RQ + 1];
  __shared__ double sE[16][RQ];
  const ETile te = tiles[blockIdx.x];   // te.k = column chunk index
  EighBlock* eb = &blocks[te.block];
  const int n = eb->n, ld = eb->npad, tid = threadIdx.x;
  const int i0 = te.k * RQ;
  if (i0 >= n) return;
  const float* D = eb->D;
  const float* V = eb->V;
  const int tr = tid >> 4, tc = tid & 15;       // thread -> rows 4*tr.., columns 4*tc..
  double e_acc[4] = {0.0, 0.0, 0.0, 0.0};
  const int npanel = (n + RQ - 1) / RQ;
  for (int J = 0; J < npanel; ++J) {
    const int j0 = J * RQ;
    double x[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) x[a][b] = 0.0;
    for (int k0 = 0; k0 < n; k0 += RK) {
      // D[j0 .. j0+63][k0 .. k0+15] and V[k0 .. k0+15][i0 .. i0+63]  (npad is a multiple of
      // 128 and the padding is zero, so no guards are needed)
      for (int e = tid; e < RQ * RK; e += 256) {
        const int r = e / RK, c = e % RK;
        sD[r][c] = gload1(D + (int64_t)(j0 + r) * ld + k0 + c);
      }
      for (int e = tid; e < RK * RQ; e += 256) {
        const int r = e / RQ, c = e % RQ;
        sV[r][c] = gload1(V + (int64_t)(k0 + r) * ld + i0 + c);
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < RK; ++k) {
        double dv[4], vv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) dv[a] = (double)sD[4 * tr + a][k];
#pragma unroll
        for (int b = 0; b < 4; ++b) vv[b] = (double)sV[k][4 * tc + b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) x[a][b] = fma(dv[a], vv[b], x[a][b]);
      }
      __syncthreads();
    }
    // e_i += sum_j V[j][i] * X[j][i] over this panel's rows
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        e_acc[b] = fma((double)gload1(V + (int64_t)(j0 + 4 * tr + a) * ld + i0 + 4 * tc + b),
                       x[a][b], e_acc[b]);
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) sE[tr][4 * tc + b] = e_acc[b];
  __syncthreads();
  if (tid < RQ && i0 + tid < n) {
    double e = 0.0;
    for (int r = 0; r < 16; ++r) e += sE[r][tid];
    eb->A[(int64_t)(i0 + tid) * ld + i0 + tid] = (float)e;
  }
}

// ---- A <- V^T D V with float64 accumulation (the transition to the finishing sweeps) ----
// The blocked two-sided sweeps accumulate absolute errors of a few eps32 * ||D|| in A, far
// above the small eigenvalues of a graded / rank-deficient-plus-ridge statistic (1e-6 ||D||),
// and the float32 re-projection V^T D V adds the same again.  With the projection accumulated
// in float64 (products of float32 numbers are exact in float64; T = D V is kept as a float32
// hi/lo pair in the temporaries X, W) every entry of A is correct to its own float32
// rounding, so the finishing sweeps work on small entries that are accurate RELATIVE to
// themselves and the eigenvectors of the small eigenvalues come out as accurately as LAPACK's
// (round-2 NumPy experiment: root error on a graded 129 x 129 input 2e-2 -> 1e-4).
// Tiles: 64 x 64 per workgroup (4 per 128 x 128 entry of the tile list), one wavefront per
// 32 x 32 quadrant on the float64 MFMA (v_mfma_f64_16x16x4_f64; 2 * 2n^3 flops per matrix).
// STAGE 0: (X, W) = hi / lo of D V.   STAGE 1: A = V^T (X + W).
// STAGE 2: X = 1.5 I - 0.5 V^T V (the Newton-Schulz factor of the final polish: with a float32
// Gram matrix the off-diagonal entries of V^T V carry sqrt(n) eps32 of noise, as large as the
// loss of orthogonality they are meant to measure).
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int STAGE>
__global__ __launch_bounds__(256) void eigh_reproject_f64_kernel(EighBlock* blocks,
                                                                 const ETile* tiles) {
  __shared__ float sL[RK][RQ + 1];
  __shared__ float sR[RK][RQ + 1];
  __shared__ float sR2[RK][RQ + 1];
  const ETile te = tiles[blockIdx.x >> 2];
  const int sub = blockIdx.x & 3;
  EighBlock* eb = &blocks[te.block];
  if (eb->small || eb->n == 0) return;
  const int ld = eb->npad, tid = threadIdx.x;
  const int i0 = te.k * TILE + (sub >> 1) * RQ, j0 = te.t * TILE + (sub & 1) * RQ;
  const float* L = STAGE == 0 ? eb->D : eb->V;
  const float* R = STAGE == 1 ? eb->X : eb->V;
  const float* R2 = eb->W;
  // v_mfma_f64_16x16x4_f64: one wavefront per 32 x 32 quadrant = 2 x 2 tiles of 16 x 16;
  // A fragment: lane -> A[i = lane & 15][k = lane >> 4], B fragment: B[k = lane >> 4][j = lane & 15],
  // accumulator element v of a lane: D[i = (lane >> 4) + 4 * v][j = lane & 15]
  const int wave = tid >> 6, lane = tid & 63;
  const int qr = 32 * (wave >> 1), qc = 32 * (wave & 1);
  const int fi = lane & 15, fk = lane >> 4;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < ld; k0 += RK) {
    {  // one 16-byte load per thread and operand (npad is a multiple of 128: always aligned)
      if (STAGE == 0) {   // left[k][m] = D[i0 + m][k0 + k]: row m holds 16 consecutive k
        const int m = tid >> 2, k4 = (tid & 3) * 4;
        const f32x4 v = gload4(L + (int64_t)(i0 + m) * ld + k0 + k4);
        sL[k4 + 0][m] = v[0]; sL[k4 + 1][m] = v[1]; sL[k4 + 2][m] = v[2]; sL[k4 + 3][m] = v[3];
      } else {            // left[k][m] = V[k0 + k][i0 + m]
        const int k = tid >> 4, m4 = (tid & 15) * 4;
        const f32x4 v = gload4(L + (int64_t)(k0 + k) * ld + i0 + m4);
        sL[k][m4 + 0] = v[0]; sL[k][m4 + 1] = v[1]; sL[k][m4 + 2] = v[2]; sL[k][m4 + 3] = v[3];
      }
      const int k = tid >> 4, c4 = (tid & 15) * 4;
      const f32x4 r = gload4(R + (int64_t)(k0 + k) * ld + j0 + c4);
      sR[k][c4 + 0] = r[0]; sR[k][c4 + 1] = r[1]; sR[k][c4 + 2] = r[2]; sR[k][c4 + 3] = r[3];
      if (STAGE == 1) {
        const f32x4 r2 = gload4(R2 + (int64_t)(k0 + k) * ld + j0 + c4);
        sR2[k][c4 + 0] = r2[0]; sR2[k][c4 + 1] = r2[1]; sR2[k][c4 + 2] = r2[2]; sR2[k][c4 + 3] = r2[3];
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < RK; kk += 4) {
      double af[2], bf[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = (double)sL[kk + fk][qr + 16 * a + fi];
#pragma unroll
      for (int b = 0; b < 2; ++b)
        bf[b] = STAGE != 1 ? (double)sR[kk + fk][qc + 16 * b + fi]
                           : (double)sR[kk + fk][qc + 16 * b + fi] +
                                 (double)sR2[kk + fk][qc + 16 * b + fi];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = i0 + qr + 16 * a + fk + 4 * v, col = j0 + qc + 16 * b + fi;
        const int64_t o = (int64_t)row * ld + col;
        const double x = acc[a][b][v];
        if (STAGE == 0) {
          const float hi = (float)x;
          eb->X[o] = hi;
          eb->W[o] = (float)(x - (double)hi);
        } else if (STAGE == 1) {
          eb->A[o] = (float)x;
        } else {
          eb->X[o] = (float)((row == col ? 1.5 : 0.0) - 0.5 * x);
        }
      }
}

// Largest scaled off-diagonal entry max_{i != j} |a_ij| / sqrt(|a_ii| |a_jj|) of the working
// matrix: the convergence measure that is meaningful for EVERY eigenvalue of a positive
// definite matrix, not only for those near ||A|| (the pivots' absolute off-norm is what the
// sweep phases stop on).  One workgroup per 128 x 128 tile, bit-pattern atomicMax per block.
__global__ __launch_bounds__(256) void eigh_scaled_off_kernel(EighBlock* blocks,
                                                              const ETile* tiles) {
  __shared__ float dr[TILE], dc[TILE];
  __shared__ unsigned red[4];
  const ETile te = tiles[blockIdx.x];
  EighBlock* eb = &blocks[te.block];
  if (eb->small || eb->n == 0) return;
  const int ld = eb->npad, n = eb->n, tid = threadIdx.x;
  const float* A = eb->A;
  const int r0 = te.k * TILE, c0 = te.t * TILE;
  if (tid < TILE) {
    dr[tid] = fmaxf(fabsf(gload1(A + (int64_t)(r0 + tid) * ld + r0 + tid)), 1e-37f);
    dc[tid] = fmaxf(fabsf(gload1(A + (int64_t)(c0 + tid) * ld + c0 + tid)), 1e-37f);
  }
  __syncthreads();
  unsigned m = 0;
  for (int e = tid; e < TILE * TILE; e += 256) {
    const int r = e >> 7, c = e & 127;
    const int row = r0 + r, col = c0 + c;
    if (row < n && col < n && row != col) {
      const float v = fabsf(gload1(A + (int64_t)row * ld + col)) * rsqrtf(dr[r]) * rsqrtf(dc[c]);
      const unsigned b = __float_as_uint(v);   // non-negative: bit order = value order; NaN on top
      m = b > m ? b : m;
    }
  }
  m = wave_max_u32(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    unsigned mm = red[0];
    for (int w = 1; w < 4; ++w) mm = red[w] > mm ? red[w] : mm;
    atomicMax(&eb->soff_bits, mm);
  }
}

#include "eigh_cj.hip.h"

// mode 1: eigenvalues = diag(A), eigenvectors = V[:, :n] (cropped to n x n).
__global__ __launch_bounds__(256) void eigh_copy_pairs_kernel(EighBlock* blocks,
                                                              const ETile* tiles) {
  const ETile te = tiles[blockIdx.x];
  EighBlock* eb = &blocks[te.block];
  if (eb->small) return;  // eigh_small_kernel wrote the (sorted) pairs itself
  const int ld = eb->npad, n = eb->n, tid = threadIdx.x;
  for (int e = tid; e < TILE * TILE; e += 256) {
    const int row = te.k * TILE + e / TILE, col = te.t * TILE + e % TILE;
    if (row < n && col < n) {
      eb->out[(int64_t)row * eb->ldo + col] = eb->V[(int64_t)row * ld + col];
      if (row == col) eb->evals_out[col] = eb->A[(int64_t)col * ld + col];
    }
  }
}

// One wavefront per block.  Besides the reference's error (DS:1022: only that field is populated there):
// Jacobi sweeps, power-iteration steps, and in PS_M_AVG_STEPS the condition number lambda_max / lambda_min of
// the regularised block from its final eigenvalues (+inf if not positive definite): the caller hands it back
// as ps_options.iters_hint at the next recompute, where a block far above eigh_td_max_cond skips the fast
// path's attempt (finish_eplan).
__global__ __launch_bounds__(64) void eigh_metrics_kernel(EighBlock* blocks, int nblocks, float* metrics) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= nblocks) return;
  EighBlock* eb = &blocks[b];
  const int n = eb->n;
  float lo = 3.0e38f, hi = -3.0e38f;
  bool nan = false;
  for (int i = lane; i < n; i += 64) {
    const float e = eb->evals[i];
    nan |= e != e;
    lo = fminf(lo, e); hi = fmaxf(hi, e);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off, 64));
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    nan |= __shfl_xor((int)nan, off, 64) != 0;
  }
  if (lane != 0) return;
  float* m = metrics + (int64_t)b * PS_METRICS_STRIDE;
  for (int i = 0; i < PS_METRICS_STRIDE; ++i) m[i] = 0.f;  // DS:1022: only the error
  m[PS_M_ERROR] = n == 0 ? 0.f : __uint_as_float(eb->err_bits);  // DS:1024-1028
  m[PS_M_TOTAL_ITERS] = (float)eb->sweeps;
  m[PS_M_POWER_ITERS] = (float)eb->power_iters;
  m[PS_M_AVG_STEPS] = n == 0 ? 0.f : (nan ? __uint_as_float(0x7fc00000u)
                                         : (lo > 0.f ? hi / lo : __uint_as_float(0x7f800000u)));
}

// What the root products do not write of `out` (they cover the npad x npad tiles of the effective part,
// zero beyond n inside them): everything for an all-padding block, and for a block whose padding_start
// leaves npad < n_full the frame of rows / columns npad .. n_full - 1 (DS:1016: val is zero there).
__global__ __launch_bounds__(256) void eigh_zero_out_kernel(EighBlock* blocks) {
  EighBlock* eb = &blocks[blockIdx.x];
  const int nf = eb->n_full, np = eb->n != 0 ? eb->npad : 0;
  if (np >= nf) return;
  for (int64_t e = blockIdx.y * 256 + threadIdx.x; e < (int64_t)nf * nf;
       e += (int64_t)gridDim.y * 256) {
    const int64_t row = e / nf, col = e % nf;
    if (row >= np || col >= np) eb->out[row * eb->ldo + col] = 0.f;
  }
}

}  // namespace psk

#include "eigh_td.hip.h"

// =============================================================================
using namespace psk;
using psh::Arena;

namespace {

// dev A/B (psh::Options::eigh_small = 0): every size goes through the blocked driver
bool small_eigh_enabled() { return psh::resolve(nullptr).eigh_small != 0; }

struct EPlan {
  int batch = 0, max_n = 0, max_nb = 0;
  std::vector<int> small_ids;     // blocks solved by eigh_small_kernel
  std::vector<int> n_eff, npad;
  std::vector<ETile> sq_tiles;    // (block, tm, tn) over npad^2
  std::vector<ETile> pair_tiles;  // (block, k)
  std::vector<ETile> row_tiles;   // (block, k, coltile)
  std::vector<ETile> col_tiles;   // (block, k, rowtile, which)
  std::vector<ETile> rq_tiles;    // (block, chunk of 64 eigenvectors) for the refinement
  std::vector<ETile> chol_tiles;  // (block, 64-row tile) of the blocks swept by eigh_cj.hip.h
  std::vector<int> big_ids;       // those blocks
  std::vector<ETile> cj_pair[2], cj_row[2];  // pair / row-tile lists of the two stream groups
  std::vector<int> cj_group_of;              // per entry of big_ids
  int cj_swept = 0;
  PiPlan pip;
  TdPlan td;                      // blocks of more than 128 rows: tridiagonalisation + divide and conquer
};

// sizing = true (ps_eigh_root_workspace_bytes: the caller's padding_start values are not known yet):
// the tridiagonalisation workspace is carved for every block of more than 128 rows at
// min(n, TD_MAXN), an upper bound of what any padding_start can make the call need.
// skip_hint (root calls; ps_options.iters_hint): the block's condition number at the previous recompute (column
// PS_M_AVG_STEPS of that call's metrics).  Far above the keep rule's bound (> 2 x eigh_td_max_cond; statistics
// move slowly) the block goes to the Jacobi solvers directly -- the same bits as after a hand-over (which starts
// those solvers from scratch), minus the time of the attempt.  NaN / 0 / absent: the attempt is made.
void finish_eplan(EPlan& pl, bool sizing = false, const float* skip_hint = nullptr, int hint_stride = 1,
                  float skip_above = 0.f) {
  pl.pip.build(pl.batch, pl.n_eff);
  if (sizing) {
    std::vector<int> ids, nn = pl.n_eff;
    for (int b : pl.big_ids)
      if (pl.n_eff[b] > SE_MAXN) { ids.push_back(b); nn[b] = std::min(nn[b], TD_MAXN); }
    if (!ids.empty()) td_make_plan(pl.td, ids, nn, pl.npad);
    return;
  }
  // blocks of 129 ... TD_MAXN rows take the tridiagonalisation; larger ones go straight to the Jacobi
  // solvers of the same call (like the blocks the fast path hands back)
  std::vector<int> ids;
  for (int b : pl.big_ids)
    if (pl.n_eff[b] > SE_MAXN && pl.n_eff[b] <= TD_MAXN &&
        !(skip_hint && skip_hint[(size_t)b * hint_stride] > skip_above))
      ids.push_back(b);
  if (!ids.empty()) td_make_plan(pl.td, ids, pl.n_eff, pl.npad);
}

void make_eplan(EPlan& pl, int batch, const int32_t* n, const int32_t* padding_start) {
  pl.batch = batch;
  pl.n_eff.resize(batch);
  pl.npad.resize(batch);
  for (int b = 0; b < batch; ++b) {
    int ne = n[b];
    if (padding_start) ne = std::max(0, std::min(ne, (int)padding_start[b]));
    pl.n_eff[b] = ne;
    pl.npad[b] = ne >= 1 ? psh::round_up(ne, TILE) : 0;
    pl.max_n = std::max(pl.max_n, ne);
    const int t = pl.npad[b] / TILE, nb = pl.npad[b] / JB, np = nb / 2;
    for (int i = 0; i < t; ++i)
      for (int j = 0; j < t; ++j) pl.sq_tiles.push_back({b, (short)i, (short)j, 0, 0});
    for (int c = 0; c < (ne + 63) / 64; ++c) pl.rq_tiles.push_back({b, (short)c, 0, 0, 0});
    if (ne >= 1 && ne <= SE_MAXN && small_eigh_enabled()) {  // LDS-resident solver, never swept
      pl.small_ids.push_back(b);
      continue;
    }
    pl.max_nb = std::max(pl.max_nb, nb);
    // the two stream groups alternate over the blocks that actually have pairs to sweep
    const int grp = np > 0 ? (pl.cj_swept++ & 1) : 0;
    pl.cj_group_of.push_back(grp);
    pl.big_ids.push_back(b);
    for (int i = 0; i < nb; ++i) pl.chol_tiles.push_back({b, (short)i, 0, 0, 0});
    for (int k = 0; k < np; ++k) {
      pl.pair_tiles.push_back({b, (short)k, 0, 0, 0});
      pl.cj_pair[grp].push_back({b, (short)k, 0, 0, 0});
      for (int c = 0; c < t; ++c) {
        pl.cj_row[grp].push_back({b, (short)k, (short)c, 0, 0});
        pl.row_tiles.push_back({b, (short)k, (short)c, 0, 0});
        pl.col_tiles.push_back({b, (short)k, (short)c, 0, 0});
        pl.col_tiles.push_back({b, (short)k, (short)c, 1, 0});
      }
    }
  }
}

struct ELayout {
  EighBlock* blocks;
  int* small_ids;
  ETile *sq, *pair, *row, *col, *rq, *chol, *cj_pair[2], *cj_row[2];
  int* big_ids;
  std::vector<float*> mat[5], Q, offp, ssq, evals;
  TdLayout td;
};

size_t ecarve(EPlan& pl, Arena& ar, ELayout* lo) {
  const int B = pl.batch;
  EighBlock* blocks = ar.take<EighBlock>(B);
  pl.pip.carve(ar, lo != nullptr);
  ETile* sq = ar.take<ETile>(pl.sq_tiles.size());
  ETile* pr = ar.take<ETile>(pl.pair_tiles.size());
  ETile* rw = ar.take<ETile>(pl.row_tiles.size());
  ETile* cl = ar.take<ETile>(pl.col_tiles.size());
  ETile* rq = ar.take<ETile>(std::max<size_t>(pl.rq_tiles.size(), 1));
  int* sid = ar.take<int>(std::max<size_t>(pl.small_ids.size(), 1));
  ETile* ch = ar.take<ETile>(std::max<size_t>(pl.chol_tiles.size(), 1));
  int* bid = ar.take<int>(std::max<size_t>(pl.big_ids.size(), 1));
  ETile* cjp[2]; ETile* cjr[2];
  for (int g = 0; g < 2; ++g) {
    cjp[g] = ar.take<ETile>(std::max<size_t>(pl.cj_pair[g].size(), 1));
    cjr[g] = ar.take<ETile>(std::max<size_t>(pl.cj_row[g].size(), 1));
  }
  if (lo) { lo->blocks = blocks; lo->sq = sq; lo->pair = pr; lo->row = rw; lo->col = cl;
            lo->rq = rq; lo->small_ids = sid; lo->chol = ch; lo->big_ids = bid;
            for (int g = 0; g < 2; ++g) { lo->cj_pair[g] = cjp[g]; lo->cj_row[g] = cjr[g]; } }
  for (int b = 0; b < B; ++b) {
    const size_t sq_e = (size_t)pl.npad[b] * pl.npad[b];
    for (int k = 0; k < 5; ++k) { float* m = ar.take<float>(sq_e); if (lo) lo->mat[k].push_back(m); }
    const int nb = pl.npad[b] / JB, np = nb / 2, t = pl.npad[b] / TILE;
    float* q = ar.take<float>((size_t)std::max(np, 1) * JP * JP);
    float* op = ar.take<float>((size_t)std::max(np * std::max(nb - 1, 1), 1));
    float* ss = ar.take<float>(std::max(t * t, 1));
    float* ev = ar.take<float>(std::max(pl.npad[b], 1));
    if (lo) { lo->Q.push_back(q); lo->offp.push_back(op); lo->ssq.push_back(ss);
              lo->evals.push_back(ev); }
  }
  td_carve(pl.td, ar, lo ? &lo->td : nullptr);
  return ar.off;
}

// One mapped status ring per host thread: a thread runs one call at a time, so calls
// on distinct (stream, workspace) pairs from different threads never share slots.
EStatus* epinned() {
  static thread_local EStatus* st = nullptr;
  if (!st && hipHostMalloc((void**)&st, 64 * sizeof(EStatus), hipHostMallocMapped) != hipSuccess)
    st = nullptr;
  return st;
}

}  // namespace

extern "C" size_t ps_eigh_root_workspace_bytes(int batch, const int32_t* n) {
  if (batch <= 0 || !n) return 0;
  EPlan pl;
  make_eplan(pl, batch, n, nullptr);
  finish_eplan(pl, true);
  Arena ar(nullptr, 0);
  return ecarve(pl, ar, nullptr) + 256;
}

// mode 0: inverse p-th root (out = val, metrics).  mode 1: plain eigenpairs
// (out = eigenvectors [n, n] in columns, evals_out = eigenvalues, Jacobi order).
static int eigh_driver(int mode, void* stream, const float* const* a, const int32_t* n,
                       const int32_t* lda, const int32_t* p, const int32_t* padding_start,
                       int batch, float ridge_epsilon, float error_tolerance,
                       int relative_matrix_epsilon, float* const* out, const int32_t* ldo,
                       float* const* evals_out, float* metrics, void* workspace,
                       size_t workspace_bytes, const ps_options* options) {
  PS_DEVICE_CHECK();
  bool bad_options = false;
  const psh::Options opt = psh::resolve(options, &bad_options);
  if (bad_options) return PS_EINVAL;
  if (batch <= 0 || !a || !n || !lda || !out || !ldo || !workspace) return PS_EINVAL;
  if (mode == 0 && (!p || !metrics)) return PS_EINVAL;
  if (mode == 1 && !evals_out) return PS_EINVAL;
  for (int b = 0; b < batch; ++b)
    if (n[b] < 1 || lda[b] < n[b] || ldo[b] < n[b] || !a[b] || !out[b] ||
        (mode == 0 && p[b] < 1) || (mode == 1 && !evals_out[b]))
      return PS_EINVAL;
  if (mode == 1) relative_matrix_epsilon = 0;
  hipStream_t st = (hipStream_t)stream;
  EPlan pl;
  make_eplan(pl, batch, n, padding_start);
  // The keep rule (a block's fast-path result stands only if positive definite with lambda_max / lambda_min
  // below the bound) applies to plain eigenpairs always and to roots under eigh_solver ACCURATE; AUTO roots
  // keep every block (the fast path is at or below a true float32 ssyevd's root error: options.h, ps_api.h).
  const bool keep_rule = !opt.eigh_td_force && (mode == 1 || opt.eigh_td_accurate) &&
                         opt.eigh_td_max_cond < 3.0e38f;
  const float keep_cond = keep_rule ? opt.eigh_td_max_cond : 0.f;
  finish_eplan(pl, false, (mode == 0 && keep_rule) ? opt.iters_hint : nullptr, opt.iters_hint_stride,
               2.f * opt.eigh_td_max_cond);
  pl.pip.set_options(opt);
  if (pl.max_n > 16384) return PS_EUNSUPPORTED;
  Arena ar(workspace, workspace_bytes);
  ELayout lo;
  ecarve(pl, ar, &lo);
  if (ar.overflow) return PS_EWORKSPACE;
  EStatus* status = epinned();
  if (!status) return PS_EINTERNAL;
  const size_t pair_lds = (size_t)(2 * JP * JLD + 2 * JB) * sizeof(float) + 2 * JB * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    PS_HIP(hipFuncSetAttribute((const void*)jacobi_pair_kernel,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_lds));
    attr_set = true;
  }

  std::vector<EighBlock> hb(batch);
  for (int b = 0; b < batch; ++b) {
    EighBlock& eb = hb[b];
    memset(&eb, 0, sizeof(eb));
    eb.a = a[b]; eb.out = out[b];
    eb.A = lo.mat[0][b]; eb.V = lo.mat[1][b]; eb.D = lo.mat[2][b]; eb.W = lo.mat[3][b];
    eb.X = lo.mat[4][b]; eb.Q = lo.Q[b]; eb.offpart = lo.offp[b];
    eb.sumsq_partial = lo.ssq[b]; eb.evals = lo.evals[b];
    eb.n = pl.n_eff[b]; eb.n_full = n[b]; eb.lda = lda[b]; eb.ldo = ldo[b];
    eb.npad = pl.npad[b]; eb.nb = pl.npad[b] / JB; eb.npairs = eb.nb / 2;
    eb.p = mode == 0 ? p[b] : 1;
    eb.alpha = (float)(-1.0 / eb.p);
    eb.evals_out = mode == 1 ? evals_out[b] : nullptr;
    eb.active = eb.n > 0 ? 1 : 0;
    eb.off_rel = 1.f;
  }
  for (int b : pl.small_ids) hb[b].small = 1;
  for (size_t i = 0; i < pl.big_ids.size(); ++i) hb[pl.big_ids[i]].cj_group = pl.cj_group_of[i];
  auto up = [&](void* d, const void* h, size_t bytes) -> int {
    return psh::upload_async(st, d, h, bytes);  // pinned staging ring: no stream synchronisation
  };
  int rc;
  if ((rc = up(lo.blocks, hb.data(), sizeof(EighBlock) * batch))) return rc;
  if ((rc = up(lo.sq, pl.sq_tiles.data(), sizeof(ETile) * pl.sq_tiles.size()))) return rc;
  if ((rc = up(lo.pair, pl.pair_tiles.data(), sizeof(ETile) * pl.pair_tiles.size()))) return rc;
  if ((rc = up(lo.row, pl.row_tiles.data(), sizeof(ETile) * pl.row_tiles.size()))) return rc;
  if ((rc = up(lo.col, pl.col_tiles.data(), sizeof(ETile) * pl.col_tiles.size()))) return rc;
  if (!pl.rq_tiles.empty() &&
      (rc = up(lo.rq, pl.rq_tiles.data(), sizeof(ETile) * pl.rq_tiles.size())))
    return rc;
  if (!pl.small_ids.empty() &&
      (rc = up(lo.small_ids, pl.small_ids.data(), sizeof(int) * pl.small_ids.size())))
    return rc;
  if (!pl.chol_tiles.empty() &&
      (rc = up(lo.chol, pl.chol_tiles.data(), sizeof(ETile) * pl.chol_tiles.size())))
    return rc;
  if (!pl.big_ids.empty() &&
      (rc = up(lo.big_ids, pl.big_ids.data(), sizeof(int) * pl.big_ids.size())))
    return rc;
  for (int g = 0; g < 2; ++g) {
    if (!pl.cj_pair[g].empty() &&
        (rc = up(lo.cj_pair[g], pl.cj_pair[g].data(), sizeof(ETile) * pl.cj_pair[g].size())))
      return rc;
    if (!pl.cj_row[g].empty() &&
        (rc = up(lo.cj_row[g], pl.cj_row[g].data(), sizeof(ETile) * pl.cj_row[g].size())))
      return rc;
  }
  const int nsq = (int)pl.sq_tiles.size();
  const int npair = (int)pl.pair_tiles.size();
  const int nrow = (int)pl.row_tiles.size(), ncol = (int)pl.col_tiles.size();
  const dim3 blk(256);

  // Power iteration (tol = error_tolerance, DS:996-1001) -> ridge -> D = A + ridge I.  Queued
  // again (on the streaming power iteration) if the resident one reports an expired wait at the
  // first host wait of the call (PiPlan::health).
  unsigned pi_expired_before = PiPlan::expired_total();
  auto enqueue_front = [&]() -> int {
    int rc2;
    if ((rc2 = up(lo.blocks, hb.data(), sizeof(EighBlock) * batch))) return rc2;
    if (relative_matrix_epsilon) {
      if ((rc2 = pl.pip.upload(st, a, lda))) return rc2;
      // the reference's power iteration is a plain mat-vec loop on the raw input (DS:996-1001)
      if ((rc2 = pl.pip.enqueue_symmetry(st, PS_SYMMETRY_VERIFY))) return rc2;
      if ((rc2 = pl.pip.enqueue(st, 100, error_tolerance))) return rc2;
    }
    hipLaunchKernelGGL(eigh_setup_kernel, dim3((batch + 255) / 256), blk, 0, st, lo.blocks,
                       pl.pip.d_blocks, batch, mode == 0 ? ridge_epsilon : 0.f, error_tolerance,
                       relative_matrix_epsilon);
    if (nsq > 0) {
      hipLaunchKernelGGL(eigh_init_kernel, dim3(nsq), blk, 0, st, lo.blocks, lo.sq, 0);
      hipLaunchKernelGGL(eigh_control_kernel, dim3(1), blk, 0, st, lo.blocks, batch, 0, 0.f, 0,
                         (EStatus*)nullptr);
    }
    PS_LAUNCH_CHECK();
    return 0;
  };
  if ((rc = enqueue_front())) return rc;

  if (nsq > 0) {
    auto sweep = [&]() {
      for (int r = 0; r < pl.max_nb - 1; ++r) {
        hipLaunchKernelGGL(jacobi_pair_kernel, dim3(npair), dim3(JT), pair_lds, st, lo.blocks,
                           lo.pair, r, 0);
        hipLaunchKernelGGL(jacobi_row_kernel, dim3(nrow), blk, 0, st, lo.blocks, lo.row, nrow, r);
        hipLaunchKernelGGL(jacobi_col_kernel, dim3(ncol), blk, 0, st, lo.blocks, lo.col, ncol, r);
      }
    };
    int gen = 0;
    auto run_phase = [&](float tol, int max_sweeps) -> int {
      for (int s = 0; s < max_sweeps; ++s) {
        EStatus* slot = &status[gen % 64];
        slot->gen = -1;
        sweep();
        hipLaunchKernelGGL(eigh_control_kernel, dim3(1), blk, 0, st, lo.blocks, batch, 1, tol,
                           gen, slot);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        e = hipStreamSynchronize(st);  // one sync per sweep (a sweep is tens of ms)
        if (e != hipSuccess) return (int)e;
        if (slot->gen != gen) return PS_EINTERNAL;
        if (opt.eigh_trace)
          fprintf(stderr, "eigh sweep %d: tol %.1e max off_rel (start of sweep) %.3e active %d\n",
                  gen, tol, slot->max_off, slot->active);
        ++gen;
        if (slot->active == 0) break;
      }
      return 0;
    };
    const bool any_big = npair > 0;
    auto enqueue_small = [&]() -> int {
    if (!pl.small_ids.empty()) {  // n <= 128: whole decomposition in one launch, no host wait
        const size_t small_lds = (size_t)(2 * SE_MAXN * SE_LD + 32 + 256) * sizeof(float);
        static bool small_attr = false;
        if (!small_attr) {
          PS_HIP(hipFuncSetAttribute((const void*)eigh_small_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds));
          small_attr = true;
        }
        const int small_refresh = opt.eigh_small_refresh;
        hipLaunchKernelGGL(eigh_small_kernel, dim3((unsigned)pl.small_ids.size()), dim3(SE_T),
                           small_lds, st, lo.blocks, lo.small_ids, small_refresh);
        PS_LAUNCH_CHECK();
      }
      return 0;
    };
    if ((rc = enqueue_small())) return rc;
    if (mode == 1 && !any_big) return PS_OK;  // the kernel wrote sorted pairs to the outputs
    // Root mode, blocks of more than 128 rows: one-sided block Jacobi on the Cholesky factor
    // (eigh_cj.hip.h).  PS_EIGH_CJ=0 restores the blocked two-sided solver for everything.
    const int cj_on = opt.eigh_cj;
    bool run_two_sided = any_big;
    // After the one-sided sweeps U is orthogonal to the sweep tolerance, so the Newton-Schulz polish
    // of the two-sided solver is off (root error unchanged to three digits on every test
    // spectrum).  The float64 Rayleigh quotients stay: |g_j|^2 drifts by ~1e-5 relative over the
    // ~300 float32 updates of a column, which is 0.1-0.2 absolute on eigenvalues of 1e4 in the
    // error metric max|u^T D u - diag(e)| (DS:1017-1021, failure threshold 0.1), and the refined
    // values are 20-30 % closer to float64 on rank-deficient-plus-ridge inputs.
    const int cj_refine = opt.eigh_cj_refine;
    const int cj_polish = opt.eigh_cj_polish;
    // Blocks of 129 ... 4096 rows: tridiagonalisation + divide and conquer (eigh_td.hip.h).  One host
    // wait at its end tells whether any block hit an iteration cap (or the resident power iteration
    // expired); such a call starts over on the Jacobi solvers below.
    // Blocks it keeps (all of them unless the keep rule applies, see keep_cond above: then those whose
    // spectrum spans less than eigh_td_max_cond -- a float32 tridiagonalisation leaves eps * ||D|| of
    // unstructured error, which the root function amplifies by ||D|| / lambda, exactly as the reference's
    // float32 ssyevd does) are marked td_done and skipped by everything between here and the common
    // finish; the others start over on the Jacobi solvers.
    bool td_all = false;   // every block of more than 128 rows is done
    bool td_any = false;
    if (opt.eigh_td && any_big && !pl.td.empty()) {
      for (int attempt = 0; attempt < 2; ++attempt) {
        if ((rc = td_run(st, pl.td, lo.td, lo.blocks, hb, opt.eigh_td_defl_eps, opt.eigh_td_stage,
                         keep_cond, opt.eigh_td_streams,
                         std::max(0, std::min(opt.eigh_td_tail, TD_TAIL)))))
          return rc;
        EStatus* slot = &status[63];
        slot->gen = -1;
        hipLaunchKernelGGL(td_status_kernel, dim3(1), dim3(1), 0, st, lo.td.nfail, slot);
        PS_LAUNCH_CHECK();
        PS_HIP(hipStreamSynchronize(st));
        if (slot->gen != 0) return PS_EINTERNAL;
        if (attempt == 0 && PiPlan::expired_total() != pi_expired_before) {
          if ((rc = enqueue_front())) return rc;   // ridge of some blocks is NaN: streaming power iteration now
          if ((rc = enqueue_small())) return rc;
          pi_expired_before = PiPlan::expired_total();   // handled here: the Jacobi path below must not start over
          continue;
        }
        const int nbig = (int)pl.big_ids.size();
        const int redo = slot->active + (nbig - (int)pl.td.ids.size());   // + the blocks above TD_MAXN rows
        if (opt.eigh_trace)
          fprintf(stderr, "eigh td: %d of %d block(s) handed to the Jacobi solvers (%d for an iteration cap / "
                          "non-finite input, the rest for their condition number)\n", redo, nbig, slot->pad_);
        td_all = redo == 0;
        td_any = redo < nbig;
        if (!td_all) {   // those blocks start over: A = D, V = I
          hipLaunchKernelGGL(eigh_init_kernel, dim3(nsq), blk, 0, st, lo.blocks, lo.sq, 1);
          PS_LAUNCH_CHECK();
        }
        break;
      }
    }
    const bool td_done = td_all;
    if (td_done) run_two_sided = false;
    if (cj_on && mode == 0 && any_big && !td_done) {
      const float cj_tol = opt.eigh_sweep_tol;
      const int cj_inner = opt.eigh_cj_inner;
      const float cj_done = opt.eigh_cj_done;   // inner iteration: stop after a sweep below this cosine
      const int cj_max_sweeps = opt.eigh_cj_max_sweeps;
      const int cj_stationary = opt.eigh_cj_stationary;
      const float cj_one_below = opt.eigh_cj_one_below;
      const int cj_sort = opt.eigh_cj_sort;
      const size_t piv_lds = (size_t)(2 * SE_MAXN * SE_LD + 32 + 3 * SE_MAXN) * sizeof(float);
      static bool piv_attr = false;
      if (!piv_attr) {
        PS_HIP(hipFuncSetAttribute((const void*)cj_pivot_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)piv_lds));
        piv_attr = true;
      }
      static thread_local hipEvent_t cj_ev[2] = {nullptr, nullptr};
      for (int i = 0; i < 2; ++i)
        if (!cj_ev[i]) PS_HIP(hipEventCreateWithFlags(&cj_ev[i], hipEventDisableTiming));
      const int nchol = (int)pl.chol_tiles.size(), nbig = (int)pl.big_ids.size();
      int gen = 0;
      for (int attempt = 0; attempt < 2; ++attempt) {
        hipLaunchKernelGGL(cj_select_kernel, dim3((batch + 255) / 256), blk, 0, st, lo.blocks, batch);
        hipLaunchKernelGGL(cj_zero_upper_kernel, dim3(nsq), blk, 0, st, lo.blocks, lo.sq);
        for (int j = 0; j < pl.max_nb; ++j) {
          hipLaunchKernelGGL(cj_chol_schur_kernel, dim3(nchol), blk, 0, st, lo.blocks, lo.chol, j);
          hipLaunchKernelGGL(cj_chol_potrf_kernel, dim3(nbig), blk, 0, st, lo.blocks, lo.big_ids, j);
          hipLaunchKernelGGL(cj_chol_trsm_kernel, dim3(nchol), blk, 0, st, lo.blocks, lo.chol, j);
        }
        EStatus* slot = &status[(2 * gen) % 64];
        slot->gen = -1;
        hipLaunchKernelGGL(cj_control_kernel, dim3(1), blk, 0, st, lo.blocks, batch, 0, 0.f, gen,
                           slot, -1);
        PS_LAUNCH_CHECK();
        PS_HIP(hipStreamSynchronize(st));   // the only stall of the path: fallback blocks known
        if (slot->gen != gen) return PS_EINTERNAL;
        ++gen;
        if (attempt == 0 && PiPlan::expired_total() != pi_expired_before) {
          // the resident power iteration gave up (see PiPlan::health): the ridge of its blocks
          // is NaN.  The process is on the streaming execution now: queue the front again.
          if ((rc = enqueue_front())) return rc;
          if ((rc = enqueue_small())) return rc;
          continue;
        }
        run_two_sided = slot->pad_ > 0;
        if (opt.eigh_trace)
          fprintf(stderr, "eigh cj: Cholesky done, %d block(s) fall back to the two-sided solver\n",
                  slot->pad_);
        break;
      }
      // Sweeps.  The blocks are dealt to two groups that sweep on two streams (the caller's and
      // a side stream of this thread): the pivot kernel is LDS / VALU work with one workgroup
      // per CU, the Gram and update kernels are MFMA work with small LDS footprints, so one
      // group's pivots run beside the other group's products on the same CUs (PS_EIGH_CJ_STREAMS=1:
      // everything on the caller's stream).  The host stays one sweep ahead of the GPU (it waits
      // for the status of sweep s - 1 only after sweep s is queued), so the streams never
      // drain; converged blocks' workgroups exit at once.
      // Two stream groups by default.  With the x6 update and x6 Gram kernels (61 KB of LDS each: neither
      // fits beside a 134 KB pivot workgroup) one stream is within +-3 % of two at 64 x 2048^2, faster on
      // one box (466 vs 476 ms) and slower on the next (500 vs 483); smaller problems gain from two
      // (64 x 1024^2: 72 vs 80 ms, 256 x 512^2: 50 vs 54 ms).
      const int nstreams = opt.eigh_streams > 0 ? opt.eigh_streams : 2;
      hipStream_t side = psh::side_stream(0);   // the library's shared pool (common.h)
      static thread_local hipEvent_t side_ev[3] = {nullptr, nullptr, nullptr};
      if (!side) return PS_EINTERNAL;
      for (int i = 0; i < 3; ++i)
        if (!side_ev[i]) PS_HIP(hipEventCreateWithFlags(&side_ev[i], hipEventDisableTiming));
      // K-tile depth of the update products: 8 = 20 KB of LDS, which fits beside a pivot workgroup
      const int cj_ubk = opt.eigh_cj_ubk;
      // arithmetic of the update products [G_I G_J] Q: 1 = three-way bf16 split on the bf16 MFMA
      const int cj_x6 = opt.eigh_update_bf16x6;
      constexpr size_t x6_lds = 6 * (size_t)XPLANE * sizeof(uint16_t);
      if (cj_x6) {
        static std::once_flag x6_once;
        std::call_once(x6_once, [] {
          (void)hipFuncSetAttribute((const void*)cj_update_x6_kernel,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)x6_lds);
        });
      }
      const bool two = nstreams >= 2 && !pl.cj_pair[1].empty();
      hipStream_t gs[2] = {st, two ? side : st};
      // Whatever way this scope is left (the error returns inside the sweep loop included), the side
      // stream is joined back into the caller's stream: the call stays stream-ordered as a whole.
      struct SideJoin {
        hipStream_t st, side; hipEvent_t ev; bool forked;
        ~SideJoin() {
          if (forked && hipEventRecord(ev, side) == hipSuccess) (void)hipStreamWaitEvent(st, ev, 0);
        }
      } side_join{st, side, side_ev[2], false};
      if (two) {   // everything queued so far on the caller's stream happens before the side stream starts
        PS_HIP(hipEventRecord(side_ev[2], st));
        PS_HIP(hipStreamWaitEvent(side, side_ev[2], 0));
        side_join.forked = true;
      }
      const int first_gen = gen;
      bool done[2] = {pl.cj_pair[0].empty(), pl.cj_pair[1].empty()};
      // Gram arithmetic per group and sweep: bf16x3 while the last KNOWN largest scaled entry of
      // the group (one sweep behind the GPU) is above cj_x3_above; float32 afterwards, and always
      // for the sweep that may stop (cj_gram_x3_kernel's comment).  OFF by default (threshold 0): a
      // conservative 5e-2 (the first 3-4 of ~10 sweeps) buys 1 % and still costs accuracy on small
      // blocks whose few sweeps outrun the one-sweep lag of `known_off` (200^2 Wishart: 3.7e-6 vs
      // 1.5e-6 from float64); with 3e-3 / 1e-3 the step is 4-7 % faster
      // (472 / 460 vs 495 ms for 64 x 2048^2) but the root ends 2 x further from float64 (7.6e-6
      // vs 4.2e-6; 1000^2: 5.9e-6 vs 2.7e-6) -- the float32 sweeps that follow a noisy phase start
      // from 1e-5-level entries, and with the clustered eigenvalues of these spectra ONE sweep
      // takes them just under the 2e-6 stop tolerance instead of far below it as the gradual
      // float32 descent does.
      const float cj_x3_above = opt.eigh_gram_x3_above;
      float known_off[2] = {1.f, 1.f};
      for (int s = 0; s < cj_max_sweeps; ++s) {
        for (int r = 0; r < pl.max_nb - 1; ++r)
          for (int g = 0; g < 2; ++g) {
            if (done[g]) continue;
            const int np_g = (int)pl.cj_pair[g].size(), nr_g = (int)pl.cj_row[g].size();
            const bool x3 = cj_x3_above > 0.f && known_off[g] > cj_x3_above;
            if (x3)
              hipLaunchKernelGGL(cj_gram_x3_kernel<2>, dim3(np_g), blk, 0, gs[g], lo.blocks,
                                 lo.cj_pair[g], np_g, r);
            else if (opt.eigh_gram_bf16x6)
              hipLaunchKernelGGL(cj_gram_x3_kernel<3>, dim3(np_g), blk, 0, gs[g], lo.blocks,
                                 lo.cj_pair[g], np_g, r);
            else
              hipLaunchKernelGGL(cj_gram_kernel, dim3(np_g), blk, 0, gs[g], lo.blocks, lo.cj_pair[g],
                                 np_g, r);
            // on a bf16 Gram matrix entries below its noise floor (~1e-5 scaled) are not rotated:
            // a rotation by noise would undo what the pair has already converged to
            hipLaunchKernelGGL(cj_pivot_kernel, dim3(np_g), dim3(SE_T), piv_lds, gs[g], lo.blocks,
                               lo.cj_pair[g], r, x3 ? fmaxf(cj_tol, opt.eigh_gram_x3_skip) : cj_tol, cj_inner, cj_done * cj_done, cj_sort,
                               cj_stationary, cj_one_below, cj_x6);
            if (cj_x6)
              hipLaunchKernelGGL(cj_update_x6_kernel, dim3(nr_g), blk, x6_lds, gs[g], lo.blocks,
                                 lo.cj_row[g], nr_g, r);
            else if (cj_ubk == 8)
              hipLaunchKernelGGL(cj_update_kernel_t<8>, dim3(nr_g), blk, 0, gs[g], lo.blocks,
                                 lo.cj_row[g], nr_g, r);
            else
              hipLaunchKernelGGL(cj_update_kernel_t<16>, dim3(nr_g), blk, 0, gs[g], lo.blocks,
                                 lo.cj_row[g], nr_g, r);
          }
        for (int g = 0; g < 2; ++g) {
          if (done[g]) continue;
          EStatus* slot = &status[(2 * gen + g) % 64];
          slot->gen = -1;
          hipLaunchKernelGGL(cj_control_kernel, dim3(1), blk, 0, gs[g], lo.blocks, batch, 1, cj_tol,
                             gen, slot, g);
          PS_HIP(hipEventRecord(g == 0 ? cj_ev[s & 1] : side_ev[s & 1], gs[g]));
        }
        PS_LAUNCH_CHECK();
        ++gen;
        if (s >= 1) {
          for (int g = 0; g < 2; ++g) {
            if (done[g]) continue;
            PS_HIP(hipEventSynchronize(g == 0 ? cj_ev[(s - 1) & 1] : side_ev[(s - 1) & 1]));
            EStatus* prev = &status[(2 * (gen - 2) + g) % 64];
            if (prev->gen != gen - 2) return PS_EINTERNAL;
            if (opt.eigh_trace)
              fprintf(stderr, "eigh cj sweep %d group %d: max scaled Gram entry %.3e, %d block(s) still sweeping\n",
                      gen - 2 - first_gen, g, prev->max_off, prev->active);
            known_off[g] = prev->max_off;
            if (prev->active == 0) done[g] = true;
          }
          if (done[0] && done[1]) break;
        }
      }
      if (two) {
        side_join.forked = false;
        PS_HIP(hipEventRecord(side_ev[2], side));
        PS_HIP(hipStreamWaitEvent(st, side_ev[2], 0));
      }
      hipLaunchKernelGGL(cj_norms_kernel, dim3(nsq), blk, 0, st, lo.blocks, lo.sq);
      hipLaunchKernelGGL(cj_finalize_kernel, dim3(nsq), blk, 0, st, lo.blocks, lo.sq);
      PS_LAUNCH_CHECK();
    }
    if (run_two_sided) {
    // phase 1: sweep until the pivot off-norm at the start of a sweep < 1e-3 ||D||
    if (any_big && (rc = run_phase(1e-3f, 30))) return rc;
    // polish: V <- V (1.5 I - 0.5 V^T V);  A <- V^T D V
    hipLaunchKernelGGL((eigh_gemm_kernel<MC, MC>), dim3(nsq), blk, 0, st, lo.blocks, lo.sq, nsq,
                       (int)GB_V, (int)GB_V, (int)GB_X, (int)GE_POLISH);
    hipLaunchKernelGGL((eigh_gemm_kernel<KC, MC>), dim3(nsq), blk, 0, st, lo.blocks, lo.sq, nsq,
                       (int)GB_V, (int)GB_X, (int)GB_W, (int)GE_STORE);
    hipLaunchKernelGGL(eigh_set_active_kernel, dim3((batch + 255) / 256), blk, 0, st, lo.blocks,
                       batch, 1);  // V <-> W
    const int f64_reproject = opt.eigh_f64_reproject;
    if (f64_reproject && any_big) {
      hipLaunchKernelGGL(eigh_reproject_f64_kernel<0>, dim3(4 * nsq), blk, 0, st, lo.blocks, lo.sq);
      hipLaunchKernelGGL(eigh_reproject_f64_kernel<1>, dim3(4 * nsq), blk, 0, st, lo.blocks, lo.sq);
    } else {
      hipLaunchKernelGGL((eigh_gemm_kernel<KC, MC>), dim3(nsq), blk, 0, st, lo.blocks, lo.sq, nsq,
                         (int)GB_D, (int)GB_V, (int)GB_X, (int)GE_STORE);
      hipLaunchKernelGGL((eigh_gemm_kernel<MC, MC>), dim3(nsq), blk, 0, st, lo.blocks, lo.sq, nsq,
                         (int)GB_V, (int)GB_X, (int)GB_A, (int)GE_STORE);
    }
    PS_LAUNCH_CHECK();
    // phase 2: finish (quadratic): stop once a sweep STARTED below 1e-4
    if (any_big && (rc = run_phase(1e-4f, 4))) return rc;
    // phase 3: the absolute off-norm says nothing about the small eigenvalues of a graded
    // spectrum: sweep on while the largest SCALED off-diagonal entry is above the float32
    // floor (well-conditioned inputs are far below it after phase 2 and skip this)
    const float scaled_tol = opt.eigh_scaled_tol;
    const int extra_sweeps = opt.eigh_extra_sweeps;
    for (int extra = 0; any_big && extra <= extra_sweeps; ++extra) {
      EStatus* slot = &status[gen % 64];
      slot->gen = -1;
      hipLaunchKernelGGL(eigh_scaled_off_kernel, dim3(nsq), blk, 0, st, lo.blocks, lo.sq);
      hipLaunchKernelGGL(eigh_control_kernel, dim3(1), blk, 0, st, lo.blocks, batch, 2, scaled_tol,
                         gen, slot);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return (int)e;
      if ((e = hipStreamSynchronize(st)) != hipSuccess) return (int)e;
      if (slot->gen != gen) return PS_EINTERNAL;
      if (opt.eigh_trace)
        fprintf(stderr, "eigh scaled off-diagonal max %.3e, %d block(s) above %.1e\n",
                slot->max_off, slot->active, scaled_tol);
      ++gen;
      if (slot->active == 0 || extra == extra_sweeps) break;
      if ((rc = run_phase(0.f, 1))) return rc;
    }
    }
    {
      const int final_polish = opt.eigh_final_polish;
      // the sweeps after the re-projection let V drift from orthogonality again by a few
      // eps32 per sweep: one more Newton-Schulz step V <- V (1.5 I - 0.5 V^T V) (two products)
      // takes most graded / rank-deficient cases to LAPACK-float32's error to three digits
      if (final_polish && any_big && (run_two_sided || cj_polish)) {
        if (final_polish >= 2)
          hipLaunchKernelGGL(eigh_reproject_f64_kernel<2>, dim3(4 * nsq), blk, 0, st, lo.blocks, lo.sq);
        else
          hipLaunchKernelGGL((eigh_gemm_kernel<MC, MC>), dim3(nsq), blk, 0, st, lo.blocks, lo.sq, nsq,
                             (int)GB_V, (int)GB_V, (int)GB_X, (int)GE_POLISH);
        hipLaunchKernelGGL((eigh_gemm_kernel<KC, MC>), dim3(nsq), blk, 0, st, lo.blocks, lo.sq, nsq,
                           (int)GB_V, (int)GB_X, (int)GB_W, (int)GE_STORE);
        hipLaunchKernelGGL(eigh_set_active_kernel, dim3((batch + 255) / 256), blk, 0, st, lo.blocks,
                           batch, 1);  // V <-> W
      }
    }
    // One-sided path only (no small blocks, no fallback): D U is formed ONCE, float64-accumulated
    // on the float64 MFMA as a float32 hi/lo pair; the Rayleigh quotients, and the error metric's
    // product U^T (D U), are taken from it (39 + 8.6 ms of float64 VALU dot products and a float32
    // D U product become 20 ms).
    if (td_any) {   // the finish of the one-sided path (Rayleigh quotients from D V) applies to them
      hipLaunchKernelGGL(td_mark_cj_kernel, dim3((batch + 255) / 256), blk, 0, st, lo.blocks, batch);
      PS_LAUNCH_CHECK();
    }
    const bool cj_only = (cj_on || td_done) && mode == 0 && any_big && !run_two_sided &&
                         pl.small_ids.empty() && cj_refine;
    if (cj_only) {
      hipLaunchKernelGGL(eigh_reproject_f64_kernel<0>, dim3(4 * nsq), blk, 0, st, lo.blocks, lo.sq);
      hipLaunchKernelGGL(cj_rayleigh_from_dv_kernel, dim3(nsq), blk, 0, st, lo.blocks, lo.sq);
      hipLaunchKernelGGL(eigh_scale_kernel, dim3(nsq), blk, 0, st, lo.blocks, lo.sq);
      hipLaunchKernelGGL((eigh_gemm_kernel<KC, KC>), dim3(nsq), blk, 0, st, lo.blocks, lo.sq, nsq,
                         (int)GB_W, (int)GB_W, (int)GB_OUT, (int)GE_SYM_STORE);
      hipLaunchKernelGGL((eigh_gemm_kernel<MC, MC>), dim3(nsq), blk, 0, st, lo.blocks, lo.sq, nsq,
                         (int)GB_V, (int)GB_X, (int)GB_A, (int)GE_ERR_UPPER);
    } else {
    {
      const int refine = opt.eigh_refine;
      if (refine && !pl.rq_tiles.empty() && (run_two_sided || cj_refine || !pl.small_ids.empty())) {
        hipLaunchKernelGGL(eigh_rayleigh_f64_kernel, dim3((unsigned)pl.rq_tiles.size()), blk, 0, st,
                           lo.blocks, lo.rq);
        PS_LAUNCH_CHECK();
      }
    }

    if (mode == 1) {
      hipLaunchKernelGGL(eigh_copy_pairs_kernel, dim3(nsq), blk, 0, st, lo.blocks, lo.sq);
      PS_LAUNCH_CHECK();
      return PS_OK;
    }
    // root and error metric
    hipLaunchKernelGGL(eigh_scale_kernel, dim3(nsq), blk, 0, st, lo.blocks, lo.sq);
    hipLaunchKernelGGL((eigh_gemm_kernel<KC, KC>), dim3(nsq), blk, 0, st, lo.blocks, lo.sq, nsq,
                       (int)GB_W, (int)GB_W, (int)GB_OUT, (int)GE_STORE);
    hipLaunchKernelGGL((eigh_gemm_kernel<KC, MC>), dim3(nsq), blk, 0, st, lo.blocks, lo.sq, nsq,
                       (int)GB_D, (int)GB_V, (int)GB_X, (int)GE_STORE);
    hipLaunchKernelGGL((eigh_gemm_kernel<MC, MC>), dim3(nsq), blk, 0, st, lo.blocks, lo.sq, nsq,
                       (int)GB_V, (int)GB_X, (int)GB_A, (int)GE_ERR);
    }
  }
  if (mode == 1) return PS_OK;
  hipLaunchKernelGGL(eigh_zero_out_kernel, dim3(batch, 16), blk, 0, st, lo.blocks);
  hipLaunchKernelGGL(eigh_metrics_kernel, dim3(batch), dim3(64), 0, st, lo.blocks, batch, metrics);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_eigh_root_batched_f32(void* stream, const float* const* a,
                                        const int32_t* n, const int32_t* lda,
                                        const int32_t* p, const int32_t* padding_start,
                                        int batch, float ridge_epsilon,
                                        float error_tolerance, int relative_matrix_epsilon,
                                        float* const* out, const int32_t* ldo,
                                        float* metrics, void* workspace,
                                        size_t workspace_bytes) {
  return eigh_driver(0, stream, a, n, lda, p, padding_start, batch, ridge_epsilon,
                     error_tolerance, relative_matrix_epsilon, out, ldo, nullptr, metrics,
                     workspace, workspace_bytes, nullptr);
}

extern "C" int ps_eigh_root_batched_opt_f32(void* stream, const float* const* a,
                                            const int32_t* n, const int32_t* lda,
                                            const int32_t* p, const int32_t* padding_start,
                                            int batch, float ridge_epsilon,
                                            float error_tolerance, int relative_matrix_epsilon,
                                            float* const* out, const int32_t* ldo,
                                            float* metrics, void* workspace,
                                            size_t workspace_bytes, const ps_options* options) {
  return eigh_driver(0, stream, a, n, lda, p, padding_start, batch, ridge_epsilon,
                     error_tolerance, relative_matrix_epsilon, out, ldo, nullptr, metrics,
                     workspace, workspace_bytes, options);
}

extern "C" int ps_eigh_sorted_max_n(void) { return small_eigh_enabled() ? SE_MAXN : 0; }

extern "C" int ps_eigh_batched_f32(void* stream, const float* const* a, const int32_t* n,
                                   const int32_t* lda, int batch, float* const* evals,
                                   float* const* evecs, const int32_t* ldv, void* workspace,
                                   size_t workspace_bytes) {
  return eigh_driver(1, stream, a, n, lda, nullptr, nullptr, batch, 0.f, 0.f, 0, evecs, ldv,
                     evals, nullptr, workspace, workspace_bytes, nullptr);
}

extern "C" int ps_eigh_batched_opt_f32(void* stream, const float* const* a, const int32_t* n,
                                       const int32_t* lda, int batch, float* const* evals,
                                       float* const* evecs, const int32_t* ldv, void* workspace,
                                       size_t workspace_bytes, const ps_options* options) {
  return eigh_driver(1, stream, a, n, lda, nullptr, nullptr, batch, 0.f, 0.f, 0, evecs, ldv,
                     evals, nullptr, workspace, workspace_bytes, options);
}
