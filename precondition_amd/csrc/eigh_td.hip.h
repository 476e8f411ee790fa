// eigh_td.hip.h — symmetric eigendecomposition by tridiagonalisation + divide and conquer
// (round 5; the algorithm class of LAPACK's ssyevd, which is what jnp.linalg.eigh runs at DS:1007
// on the reference's CPU path).  Replaces the block Jacobi sweeps (~60 n^3 executed flops, 40 % of
// the time in a VALU pivot kernel) for matrices of more than 128 rows:
//
//   D = Q T Q^T     Householder reduction to tridiagonal form, panels of TD_NB columns:
//                   per column three short launches over the whole batch (row update + norm,
//                   symmetric mat-vec on the upper 128 x 128 tiles, w vector), HBM-bound on the
//                   mat-vec (the trailing triangle is read once per column: n^3 / 6 * 4 bytes);
//                   the rank-2k update of the trailing matrix once per panel on the fp32 MFMA;
//   T = Z_T L Z_T^T float64 divide and conquer (dc_core.h: QL leaves of <= 32 rows, deflation, secular
//                   equation, Loewner weights), merges as grouped products [Q1 S_top; Q2 S_bot] on
//                   the fp32 MFMA (4/3 n^3);
//   Z = Q Z_T       blocked compact-WY back-transformation, 128 reflectors per block:
//                   T^-1 = diag(1 / tau) + striu(V^T V), Y = (T V^T) Z, Z -= V Y (2 n^3, fp32 MFMA).
//
// All reductions are fixed-order (no float atomics): results are bit-reproducible.
// Layouts: reflector j is ROW j of VHt (so the rows-parallel kernels read it coalesced and the MFMA
// operands need no transposition); the panel's W vectors are rows of Wt[TD_NB][ld].
#pragma once
#include "dc_core.h"
#include "dc_plan.h"

namespace psk {

constexpr int TD_NB = 32;     // panel width of the reduction
constexpr int TD_KB = 128;    // reflectors per WY block of the back-transformation
constexpr int TD_MAXN = 4096; // the deflation kernel keeps a merge problem in LDS
constexpr int TD_TAIL = 192;  // the last <= 192 columns of a block are reduced inside LDS by one workgroup
static_assert(TD_TAIL <= 192, "td_tail_kernel: one thread per row, three 64-column strips, 160 KB of LDS");

struct TdBlock {
  int eb;              // index of the EighBlock
  int n, ld, nt;       // size, leading dimension (multiple of 128), 128-row tiles
  int height;          // merge levels of the partition tree
  int fail;            // QL / secular iteration caps hit, or NaN input: solved again by the Jacobi path
  int keep;            // decided after the divide and conquer: this block's result stands
  int jtail;           // columns jtail .. n - 1 are reduced by td_tail_kernel (n: no tail); a multiple of TD_NB
  float* A;            // working matrix: upper 128 x 128 tiles (I <= J) are maintained
  float* VHt;          // [ld][ld] reflector j in row j (zero up to and including column j)
  float* Wt;           // [TD_NB][ld]
  float* ubuf;         // [ld] updated row of the current column
  float* wp;           // [ld] w before its last correction
  float* slab;         // [nt][nt][128] partial mat-vec products
  double* part_ss;     // [nt] float64: squares of a trailing matrix that is rounding noise underflow in float32
  float* part_dot;     // [nt]
  float* part_ab;      // [nt][2][TD_NB]
  float* dT; float* eT; float* tau;   // [ld]
  // divide and conquer (float64 arrays of length ld, indexed by the global row of the node)
  double* d64; double* e64; double* Dc[2];
  double* z; double* dl; double* w; double* mu; double* zh; double* rn; double* dflv; double* rc; double* rs;
  int* perm; int* col; int* dflc; int* ra; int* rb; int* org; int* rowmap; int* colkind;
  double scale;
  float* Q[2];         // eigenvector matrices of the current / next level
  float* S;            // secular vectors of the level; afterwards V2 = T V^T of the back-transformation
  float* TT;           // [nbk][128 * 128] T^-1, then T, of every WY block
  float* Y2;           // [128][ld]
  float* Z;            // where the eigenvectors end up (Q[height & 1 ? ...])
  float* evals;        // [ld] float32 eigenvalues (ascending), 0 beyond n
};

struct TdNode {
  int blk, r0, m, n1;
};
struct TdNodeSt {
  int K, ndefl, nrot, pad_;
  double rho;
};
struct TdGTile {          // one 128 x 128 output tile of a merge product
  int blk, r0, m, r0h, nh;
  short tm, tn;
};

__device__ __forceinline__ float td_wg_sum128(float v, float* red) {   // 128 threads
  v = wave_sum_f32(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = red[0] + red[1];
  __syncthreads();
  return r;
}
__device__ __forceinline__ double td_wg_sum128_f64(double v, double* red) {   // 128 threads
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = red[0] + red[1];
  __syncthreads();
  return r;
}

// Sum of p[x], x0 <= x < nt (nt <= 32), the same bits in every lane of the calling wavefront: lane x requests
// p[x] (ONE coalesced load; a loop over the uniform addresses compiles to up to 32 dependent loads with the
// pointer re-read in between -- tools/bench_symv.hip v0d: a third of the mat-vec's time), then a butterfly.
__device__ __forceinline__ double td_sum32_f64(const double* p, int x0, int nt) {
  const int l = threadIdx.x & 63;
  double v = (l >= x0 && l < nt) ? p[l] : 0.0;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v + __shfl_xor(v, 32, 64);   // lanes 32 .. 63 held 0
}
__device__ __forceinline__ float td_sum32_f32(const float* p, int x0, int nt) {
  const int l = threadIdx.x & 63;
  float v = (l >= x0 && l < nt) ? p[l] : 0.f;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ---- reduction, per column: (1) finish w of the previous column, updated row j, its norm ----------
// grid (nt_max - j / 128, nblk), 128 threads: tile row X = j / 128 + blockIdx.x.
//   jf >= 0: W[:, jf] = wp - gamma v_jf with gamma = tau_jf / 2 * (wp . v_jf)
//   jr >= 0: u[c] = A[jr][c] - sum_{i' < i} (V[jr, i'] W[c, i'] + W[jr, i'] V[c, i']),  d[jr] = u[jr],
//            partial sums of u[c]^2 over c >= jr + 2
__global__ __launch_bounds__(128) void td_row_kernel(TdBlock* blocks, int jf, int jr) {
  __shared__ double red[2];
  __shared__ float sVj[TD_NB], sWj[TD_NB];
  // the descriptor by value: ONE burst of scalar loads (through the pointer every field access is a
  // dependent s_load + wait: the kernel writes global memory, so nothing read through it is hoisted)
  const TdBlock tbv = blocks[blockIdx.y];
  const TdBlock* tb = &tbv;
  const int n = tb->n, ld = tb->ld, nt = tb->nt, tid = threadIdx.x;
  const int jbase = jf >= 0 ? jf + 1 : jr;
  const int X = jbase / TILE + blockIdx.x;
  if (X >= nt) return;
  const int c = X * TILE + tid;
  const bool do_f = jf >= 0 && jf <= n - 2 && jf < tb->jtail;
  const bool do_r = jr >= 0 && jr <= n - 1 && jr < tb->jtail;
  const int i = do_r ? jr % TD_NB : 0, p = jr - i;
  // every load of the kernel is issued before the first use (one memory round trip)
  float pdl = 0.f;
  float tauf = 0.f, wpc = 0.f, vfc = 0.f, wpj = 0.f, vfj = 0.f;
  if (do_f) {
    const int x0 = (jf + 1) / TILE, l = tid & 63;
    pdl = (l >= x0 && l < nt) ? tb->part_dot[l] : 0.f;   // lane x holds partial x (td_sum32_f32's load, issued here)
    tauf = tb->tau[jf];
    wpc = tb->wp[c];
    vfc = tb->VHt[(int64_t)jf * ld + c];
    if (do_r) { wpj = tb->wp[jr]; vfj = tb->VHt[(int64_t)jf * ld + jr]; }
  }
  float vj = 0.f, wj = 0.f, u = 0.f;
  float wrow[TD_NB], vrow[TD_NB];
  if (do_r) {
    if (tid < i) {
      vj = tb->VHt[(int64_t)(p + tid) * ld + jr];
      wj = tb->Wt[(int64_t)tid * ld + jr];
    }
    u = tb->A[(int64_t)jr * ld + c];
#pragma unroll
    for (int ip = 0; ip < TD_NB; ++ip) {
      wrow[ip] = ip < i ? tb->Wt[(int64_t)ip * ld + c] : 0.f;
      vrow[ip] = ip < i ? tb->VHt[(int64_t)(p + ip) * ld + c] : 0.f;
    }
  }
  float wfin = 0.f, gamma = 0.f;
  if (do_f) {
    float s = pdl;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    s += __shfl_xor(s, 32, 64);
    gamma = 0.5f * tauf * s;
    wfin = wpc - gamma * vfc;
    tb->Wt[(int64_t)(jf % TD_NB) * ld + c] = wfin;
  }
  if (!do_r) return;
  if (tid < i) {
    if (do_f && tid == i - 1) wj = wpj - gamma * vfj;   // being finalised by another workgroup: recompute
    sVj[tid] = vj;
    sWj[tid] = wj;
  }
  __syncthreads();
#pragma unroll
  for (int ip = 0; ip < TD_NB; ++ip) {
    if (ip < i) {
      const float wc = (do_f && ip == i - 1) ? wfin : wrow[ip];
      u -= sVj[ip] * wc + sWj[ip] * vrow[ip];
    }
  }
  if (c >= n) u = 0.f;
  if (c == jr) tb->dT[jr] = u;
  tb->ubuf[c] = c > jr ? u : 0.f;
  // float64: the trailing matrix of an exactly low-rank input (all ones, say) is rounding noise of
  // rounding noise, 1e-8 smaller with every column, and its squares leave the float32 range (a
  // reflector with tau v^T v != 2: Q lost orthogonality to 4e-4); 1e18-sized entries overflow there
  const double ud = c >= jr + 2 ? (double)u : 0.0;
  const double ss = td_wg_sum128_f64(ud * ud, red);
  if (tid == 0) tb->part_ss[X] = ss;
}

// Householder scalars of column j from the row kernel's outputs.  tau = 0 (nothing to annihilate)
// stores the ZERO vector: the WY factor of the back-transformation then sees an identity.
__device__ __forceinline__ void td_house_scalars(float alpha, double sigma, float& beta, float& tau,
                                                 float& scale, float& v1) {
  const float sf = (float)sigma, a2 = alpha * alpha;
  if (sf > 1e-30f && sf < 1e30f && a2 < 1e30f) {   // the common case: float32 arithmetic is safe (wave-uniform branch)
    beta = -copysignf(sqrtf(a2 + sf), alpha);
    tau = (beta - alpha) / beta;
    scale = 1.f / (alpha - beta);
    v1 = 1.f;
    return;
  }
  const double ad = (double)alpha, nrm = sqrt(ad * ad + sigma);
  // sigma == 0: nothing to annihilate; a column below 1e-36 is dropped (1 / (alpha - beta) must stay a float)
  if ((!(sigma > 0.0) && sigma == sigma) || nrm < 1e-36) {
    beta = alpha; tau = 0.f; scale = 0.f; v1 = 0.f;
  } else {
    const double bd = alpha >= 0.f ? -nrm : nrm;
    beta = (float)bd;
    tau = (float)((bd - ad) / bd);
    scale = (float)(1.0 / (ad - bd));
    v1 = 1.f;
  }
}
__device__ __forceinline__ void td_house(const TdBlock* tb, int j, float& beta, float& tau,
                                         float& scale, float& v1) {
  const float alpha = tb->ubuf[j + 1];
  const double sigma = td_sum32_f64(tb->part_ss, j / TILE, tb->nt);
  td_house_scalars(alpha, sigma, beta, tau, scale, v1);
}

// ---- reduction, per column: (2) y = A v on the upper tiles; v, e_j, tau_j stored -------------------
// grid (T (T + 1) / 2, nblk) with T = nt_max - (j + 1) / 128, 256 threads: one 128 x 128 tile
// (I <= J) per workgroup; an off-diagonal tile serves y_I (row sums) and y_J (column sums).
__global__ __launch_bounds__(256) void td_symv_kernel(TdBlock* blocks, int j, int Tmax) {
  __shared__ float svI[TILE], svJ[TILE];
  __shared__ float srow[TILE];
  __shared__ float scol[8][TILE];
  // the descriptor by value: ONE burst of scalar loads (through the pointer every field access is a
  // dependent s_load + wait: the kernel writes global memory, so nothing read through it is hoisted)
  const TdBlock tbv = blocks[blockIdx.y];
  const TdBlock* tb = &tbv;
  const int n = tb->n, ld = tb->ld, nt = tb->nt, tid = threadIdx.x;
  if (j > n - 2 || j >= tb->jtail) return;
  const int I0 = (j + 1) / TILE;
  int q = blockIdx.x, Ip = 0;
  while (q >= Tmax - Ip) { q -= Tmax - Ip; ++Ip; }
  const int I = I0 + Ip, J = I + q;
  if (J >= nt) return;
  // the tile: thread -> rows (tid >> 5) + 8 k, columns 4 (tid & 31) .. ; its loads go out first
  const int r0 = tid >> 5, c4 = (tid & 31) * 4;
  // (wave-uniform 64-bit base in SGPRs) + (one 32-bit byte offset per lane): sixteen loads in flight on
  // one address register (per-load 64-bit address pairs made the allocator reuse a destination as an
  // address and wait for it: a vmcnt(0) after the second load)
  const float* ubase = tb->A + (int64_t)(I * TILE) * ld + J * TILE;
  uint32_t loff = (uint32_t)((r0 * ld + c4) * 4);
  asm volatile("" : "+v"(loff));
  f32x4 a[16];
  // rows up to j (first tile row only) carry v = 0 and their sums are not used: they are not read
  // (on average half of the first tile row: ~1 / (T + 1) of the launch's bytes)
  // The test is WAVE-UNIFORM (a wavefront holds the row pair 2 w, 2 w + 1 of every group of 8: the load is
  // skipped when both are masked): with a per-lane test every load sits under its own exec mask and the
  // compiler put a vmcnt(0) between the first ones.
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jl = j - I * TILE;               // last masked row of this tile (< 0: none)
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    a[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (8 * k + 2 * wv + 1 > jl)
      a[k] = *(const f32x4 PS_GLOBAL*)((const char PS_GLOBAL*)(ubase + (int64_t)(8 * k) * ld) + (uint64_t)loff);
  }
  // Householder scalars: wavefronts 0 and 1 only (they fill the vectors; both compute the same bits); the
  // vector's loads go out before the scalars so that everything after the tile is ONE round trip
  float beta = 0.f, tau = 0.f;
  if (wv < 2) {
    const int cI = I * TILE + tid, cJ = J * TILE + tid;
    const float uI = tb->ubuf[cI], uJ = tb->ubuf[cJ];
    float scale, v1;
    td_house(tb, j, beta, tau, scale, v1);
    svI[tid] = cI == j + 1 ? v1 : (cI > j + 1 && cI < n ? uI * scale : 0.f);
    svJ[tid] = cJ == j + 1 ? v1 : (cJ > j + 1 && cJ < n ? uJ * scale : 0.f);
  }
  __syncthreads();
  const f32x4 vj = *reinterpret_cast<const f32x4*>(&svJ[c4]);
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float rs = a[k][0] * vj[0] + a[k][1] * vj[1] + a[k][2] * vj[2] + a[k][3] * vj[3];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
    if ((tid & 31) == 0) srow[r0 + 8 * k] = rs;
    const float vi = svI[r0 + 8 * k];
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[e] += a[k][e] * vi;
  }
  if (I != J) {
#pragma unroll
    for (int e = 0; e < 4; ++e) scol[r0][c4 + e] = cs[e];
  }
  __syncthreads();
  if (tid < TILE) {
    tb->slab[((int64_t)I * nt + J) * TILE + tid] = srow[tid];
    if (I != J) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) s += scol[g][tid];
      tb->slab[((int64_t)J * nt + I) * TILE + tid] = s;
    }
  }
  if (I != J) return;
  // diagonal tile: v_j, and the partial products W^T v, V^T v of the panel for these 128 rows
  const int i = j % TD_NB, p = j - i;
  if (tid < TILE) tb->VHt[(int64_t)j * ld + I * TILE + tid] = svI[tid];
  if (I == I0 && tid == 0) { tb->eT[j] = beta; tb->tau[j] = tau; }
  {
    const int ip = tid >> 3, sub = tid & 7;
    float sa = 0.f, sb = 0.f;
    if (ip < i) {
      const float* wrow = tb->Wt + (int64_t)ip * ld + I * TILE + sub * 16;
      const float* vrow = tb->VHt + (int64_t)(p + ip) * ld + I * TILE + sub * 16;
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const f32x4 wv = gload4(wrow + 4 * e4), vv = gload4(vrow + 4 * e4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = svI[sub * 16 + 4 * e4 + e];
          sa += wv[e] * x;
          sb += vv[e] * x;
        }
      }
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) { sa += __shfl_xor(sa, off, 64); sb += __shfl_xor(sb, off, 64); }
    if (sub == 0 && ip < i) {
      tb->part_ab[(0 * TD_NB + ip) * 32 + I] = sa;   // [which][ip][tile row]: the reader's run is contiguous
      tb->part_ab[(1 * TD_NB + ip) * 32 + I] = sb;
    }
  }
}

// ---- reduction, per column: (3) w' = tau (y - V (W^T v) - W (V^T v)), partial w' . v ---------------
// grid (nt_max - (j + 1) / 128, nblk), 128 threads.
__global__ __launch_bounds__(128) void td_w_kernel(TdBlock* blocks, int j) {
  __shared__ float red[2];
  __shared__ float sab[2][TD_NB];
  // the descriptor by value: ONE burst of scalar loads (through the pointer every field access is a
  // dependent s_load + wait: the kernel writes global memory, so nothing read through it is hoisted)
  const TdBlock tbv = blocks[blockIdx.y];
  const TdBlock* tb = &tbv;
  const int n = tb->n, ld = tb->ld, nt = tb->nt, tid = threadIdx.x;
  if (j > n - 2 || j >= tb->jtail) return;
  const int I0 = (j + 1) / TILE, X = I0 + blockIdx.x;
  if (X >= nt) return;
  const int i = j % TD_NB, p = j - i;
  const int c = X * TILE + tid;
  // every load of the kernel is issued before the first use
  float pab[32];
  if (tid < 2 * TD_NB) {
    const float* src = tb->part_ab + (int64_t)tid * 32;   // tid = which * TD_NB + ip
#pragma unroll
    for (int x4 = 0; x4 < 8; ++x4) {
      const f32x4 v = gload4(src + 4 * x4);
#pragma unroll
      for (int e = 0; e < 4; ++e) pab[4 * x4 + e] = v[e];
    }
  }
  float ys[32], wrow[TD_NB], vrow[TD_NB];
#pragma unroll
  for (int Y = 0; Y < 32; ++Y)
    ys[Y] = (Y >= I0 && Y < nt) ? tb->slab[((int64_t)X * nt + Y) * TILE + tid] : 0.f;
#pragma unroll
  for (int ip = 0; ip < TD_NB; ++ip) {
    vrow[ip] = ip < i ? tb->VHt[(int64_t)(p + ip) * ld + c] : 0.f;
    wrow[ip] = ip < i ? tb->Wt[(int64_t)ip * ld + c] : 0.f;
  }
  const float tau = tb->tau[j];
  const float vc = tb->VHt[(int64_t)j * ld + c];
  if (tid < 2 * TD_NB) {
    float s = 0.f;
#pragma unroll
    for (int x = 0; x < 32; ++x) s += (x >= I0 && x < nt) ? pab[x] : 0.f;
    sab[tid / TD_NB][tid % TD_NB] = (tid % TD_NB) < i ? s : 0.f;
  }
  __syncthreads();
  float y = 0.f;
#pragma unroll
  for (int Y = 0; Y < 32; ++Y) y += ys[Y];
#pragma unroll
  for (int ip = 0; ip < TD_NB; ++ip) y -= vrow[ip] * sab[0][ip] + wrow[ip] * sab[1][ip];
  const float w = (c > j && c < n) ? tau * y : 0.f;
  tb->wp[c] = w;
  const float dot = td_wg_sum128(w * vc, red);
  if (tid == 0) tb->part_dot[X] = dot;
}

// ---- reduction, per panel: A -= V W^T + W V^T on the upper tiles of the trailing matrix -------------
// grid (T (T + 1) / 2, nblk) with T = nt_max - (p + TD_NB) / 128, 256 threads.
__global__ __launch_bounds__(256, 2) void td_syr2k_kernel(TdBlock* blocks, int p, int Tmax) {
  __shared__ __align__(16) float smem[SmemCfg<16>::TOTAL];
  const TdBlock tbv = blocks[blockIdx.y];   // by value: one burst of scalar loads
  const TdBlock* tb = &tbv;
  const int n = tb->n, ld = tb->ld, nt = tb->nt;
  if (p + TD_NB > n - 1 || p >= tb->jtail) return;   // nothing left to update
  const int I0 = (p + TD_NB) / TILE;
  int q = blockIdx.x, Ip = 0;
  while (q >= Tmax - Ip) { q -= Tmax - Ip; ++Ip; }
  const int I = I0 + Ip, J = I + q;
  if (J >= nt) return;
  const float* Vp = tb->VHt + (int64_t)p * ld;
  Operand Av{Vp, ld, I * TILE, ld, TD_NB, true}, Bw{tb->Wt, ld, J * TILE, ld, TD_NB, true};
  Operand Aw{tb->Wt, ld, I * TILE, ld, TD_NB, true}, Bv{Vp, ld, J * TILE, ld, TD_NB, true};
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1;
  // the tile of A is requested BEFORE the products (the kernel is a read-modify-write of the trailing
  // matrix with 64 flops per element: its time is the latency chain operands -> LDS -> MFMA -> tile)
  f32x16 old[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = I * TILE + acc_row(wm, a, r, lane), col = J * TILE + acc_col(wn, b, lane);
        old[a][b][r] = gload1(tb->A + (int64_t)row * ld + col);
      }
  f32x16 acc[2][2];
  zero_acc(acc);
  gemm_tile_accum<MC, MC, 16, false>(Av, Bw, TD_NB, smem, acc);
  gemm_tile_accum<MC, MC, 16, false>(Aw, Bv, TD_NB, smem, acc);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = I * TILE + acc_row(wm, a, r, lane), col = J * TILE + acc_col(wn, b, lane);
        gstore1(tb->A + (int64_t)row * ld + col, old[a][b][r] - acc[a][b][r]);
      }
}

// ---- reduction, tail: the last m = n - jtail <= TD_TAIL columns of a block inside LDS ----------------------
// Down there a column of the streaming path is three launches of ~6 us for a few KB of matrix.  One
// workgroup per block holds the (fully updated) trailing matrix in LDS and runs the classical unblocked
// reduction on it: per column the Householder vector, y = S v, w = tau (y - tau/2 (y.v) v), S -= v w^T + w v^T
// (both triangles, so that rows stay readable), ~1.2 us per column on average.  Outputs as the column kernels':
// d, e, tau, reflector k in row jtail + k of VHt.  grid (nblk), 512 threads (two per row), LDS (m (m | 1) + 2 m + 16) floats.
__global__ __launch_bounds__(512) void td_tail_kernel(TdBlock* blocks) {
  extern __shared__ __align__(16) float td_tail_lds[];
  __shared__ double s_red64[8];
  __shared__ float s_red32[8];
  const TdBlock tbv = blocks[blockIdx.x];
  const TdBlock* tb = &tbv;
  const int n = tb->n, ld = tb->ld, jt = tb->jtail, tid = threadIdx.x;
  const int m = n - jt;
  if (m <= 0) return;
  const int LS = m | 1;                  // odd row stride: a walk down a column is conflict-free
  float* S = td_tail_lds;
  float* vv = S + (size_t)m * LS;
  float* ww = vv + m;
  const int lane = tid & 63, wave = tid >> 6;   // 8 wavefronts
  // upper tiles (I <= J) of A are maintained: element (r, c) of a lower tile is read at (c, r)
  for (int r0 = 0; r0 < m; r0 += 32) {   // 4 rows x 3 column strips per wavefront and trip: 12 loads in flight
    float x[4][3];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int r = r0 + 8 * a + wave, c = lane + 64 * b;
        const int gr = jt + r, gc = jt + c;
        x[a][b] = (r < m && c < m) ? ((gr / TILE <= gc / TILE) ? tb->A[(int64_t)gr * ld + gc] : tb->A[(int64_t)gc * ld + gr]) : 0.f;
      }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int r = r0 + 8 * a + wave, c = lane + 64 * b;
        if (r < m && c < m) S[r * LS + c] = x[a][b];
      }
  }
  __syncthreads();
  // two threads per row: thread (row, h) owns the columns c = k + 1 + h, k + 3 + h, ... of its row
  const int row = tid >> 1, h = tid & 1;
  for (int k = 0; k < m; ++k) {
    if (tid == 0) tb->dT[jt + k] = S[k * LS + k];
    if (k == m - 1) break;
    // norm of the part to annihilate (float64: see the row kernel), alpha
    double ss = 0.0;
    for (int c = k + 2 + tid; c < m; c += 512) { const double u = (double)S[k * LS + c]; ss += u * u; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (lane == 0) s_red64[wave] = ss;
    __syncthreads();
    double sigma = 0.0;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) sigma += s_red64[w8];
    const float alpha = S[k * LS + k + 1];
    float beta, tau, scale, v1;
    td_house_scalars(alpha, sigma, beta, tau, scale, v1);
    for (int c = tid; c < m; c += 512) {
      const float v = c == k + 1 ? v1 : (c > k + 1 ? S[k * LS + c] * scale : 0.f);
      vv[c] = v;
      tb->VHt[(int64_t)(jt + k) * ld + jt + c] = v;
    }
    if (tid == 0) { tb->eT[jt + k] = beta; tb->tau[jt + k] = tau; }
    __syncthreads();
    if (tau != 0.f) {   // uniform
      const bool mine = row > k && row < m;
      float* sr = S + row * LS;
      // y = S v over the trailing part (v is zero up to k); eight elements per trip, loads before uses
      float y = 0.f;
      if (mine) {
        int c = k + 1 + h;
        for (; c + 14 < m; c += 16) {
          float a8[8], b8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { a8[e] = sr[c + 2 * e]; b8[e] = vv[c + 2 * e]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) y += a8[e] * b8[e];
        }
        for (; c < m; c += 2) y += sr[c] * vv[c];
      }
      y += __shfl_xor(y, 1, 64);   // the two halves of the row
      float dot = (mine && h == 0) ? y * vv[row] : 0.f;
      dot = wave_sum_f32(dot);
      if (lane == 0) s_red32[wave] = dot;
      __syncthreads();
      dot = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) dot += s_red32[w8];
      if (h == 0 && row < m) ww[row] = mine ? tau * (y - 0.5f * tau * dot * vv[row]) : 0.f;
      __syncthreads();
      if (mine) {
        const float vr = vv[row], wr = ww[row];
        int c = k + 1 + h;
        for (; c + 14 < m; c += 16) {
          float a8[8], b8[8], c8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { a8[e] = sr[c + 2 * e]; b8[e] = ww[c + 2 * e]; c8[e] = vv[c + 2 * e]; }
#pragma unroll
          for (int e = 0; e < 8; ++e) sr[c + 2 * e] = a8[e] - (vr * b8[e] + wr * c8[e]);
        }
        for (; c < m; c += 2) sr[c] -= vr * ww[c] + wr * vv[c];
      }
      __syncthreads();
    }
  }
}

// ---- zero fill of a [rows][ld] float region per block ------------------------------------------------
__global__ __launch_bounds__(256) void td_zero_kernel(TdBlock* blocks, int which) {
  TdBlock* tb = &blocks[blockIdx.y];
  const int ld = tb->ld;
  float* dst = which == 0 ? tb->VHt : tb->Wt;
  const int64_t total = which == 0 ? (int64_t)ld * ld : (int64_t)TD_NB * ld;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; e < total; e += (int64_t)gridDim.x * 1024)
    *(f32x4 PS_GLOBAL*)(dst + e) = z;
}

// ======================= divide and conquer on the tridiagonal matrix ===============================

// scale to unit max-norm (as sstedc does), float64 copies; NaN / Inf input marks the block failed
__global__ __launch_bounds__(256) void td_dc_init_kernel(TdBlock* blocks) {
  __shared__ float red[4];
  TdBlock* tb = &blocks[blockIdx.x];
  const int n = tb->n, tid = threadIdx.x;
  float mx = 0.f;
  int bad = 0;
  for (int i = tid; i < n; i += 256) {
    const float d = tb->dT[i], e = i < n - 1 ? tb->eT[i] : 0.f;
    if (!(fabsf(d) <= 3.0e38f) || !(fabsf(e) <= 3.0e38f)) bad = 1;
    mx = fmaxf(mx, fmaxf(fabsf(d), fabsf(e)));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  bad = __syncthreads_or(bad);
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const double scale = mx > 0.f ? (double)mx : 1.0;
  if (tid == 0) { tb->scale = scale; tb->fail = bad ? 1 : 0; }
  for (int i = tid; i < n; i += 256) {
    tb->d64[i] = bad ? (double)(i + 1) : (double)tb->dT[i] / scale;
    tb->e64[i] = (bad || i >= n - 1) ? 0.0 : (double)tb->eT[i] / scale;
  }
}

// T = blockdiag(T1', T2') + |beta| u u^T at every cut: the diagonal entries next to a cut lose |beta|
__global__ void td_dc_cuts_kernel(TdBlock* blocks, const TdNode* nodes, int nnodes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnodes) return;
  const TdNode nd = nodes[i];
  TdBlock* tb = &blocks[nd.blk];
  const int k = nd.r0 + nd.n1 - 1;
  const double b = fabs(tb->e64[k]);
  // cuts are at least 16 rows apart: no two threads touch the same entry
  tb->d64[k] -= b;
  tb->d64[k + 1] -= b;
}

// leaves: one wavefront per leaf, QL iteration with the tridiagonal in LDS, lane = row of Z
__global__ __launch_bounds__(256) void td_dc_leaf_kernel(TdBlock* blocks, const TdNode* leaves,
                                                        int nleaves) {
  constexpr int L = psdc::DC_LEAF, LZ = L + 1;
  __shared__ double sd[4][L], se[4][L];
  __shared__ double sz[4][L * LZ];
  __shared__ int srank[4][L];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = blockIdx.x * 4 + wave;
  if (li >= nleaves) return;
  const TdNode nd = leaves[li];
  TdBlock* tb = &blocks[nd.blk];
  const int m = nd.m, r0 = nd.r0, ld = tb->ld;
  volatile double* d = sd[wave];
  volatile double* e = se[wave];
  double* z = sz[wave];
  if (lane < m) {
    d[lane] = tb->d64[r0 + lane];
    e[lane] = lane < m - 1 ? tb->e64[r0 + lane] : 0.0;
    for (int c = 0; c < m; ++c) z[lane * LZ + c] = c == lane ? 1.0 : 0.0;
  }
  const int bad = psdc::dc_tql2<volatile double, double>(m, d, e, z, LZ, lane, 64, lane == 0);
  if (bad && lane == 0) tb->fail = 1;
  if (lane < m) {
    const double mine = d[lane];
    int rk = 0;
    for (int c = 0; c < m; ++c) {
      const double o = d[c];
      rk += (o < mine || (o == mine && c < lane)) ? 1 : 0;
    }
    srank[wave][lane] = rk;
    tb->Dc[0][r0 + rk] = mine;
  }
  if (lane < m) {
    float* qrow = tb->Q[0] + (int64_t)(r0 + lane) * ld + r0;
    for (int c = 0; c < m; ++c) qrow[srank[wave][c]] = (float)z[lane * LZ + c];
  }
}

// deflation: one workgroup per merge node; the scan itself is sequential (thread 0, data in LDS)
__global__ __launch_bounds__(256) void td_dc_deflate_kernel(TdBlock* blocks, const TdNode* nodes,
                                                           TdNodeSt* st, int src, float eps_defl) {
  extern __shared__ __align__(16) unsigned char td_dyn[];
  __shared__ double sred[2][4];
  const TdNode nd = nodes[blockIdx.x];
  TdBlock* tb = &blocks[nd.blk];
  const int m = nd.m, n1 = nd.n1, r0 = nd.r0, ld = tb->ld, tid = threadIdx.x;
  double* d = reinterpret_cast<double*>(td_dyn);
  double* z = d + m;
  int* perm = reinterpret_cast<int*>(z + m);
  const double beta = tb->e64[r0 + n1 - 1];
  const double rho = 2.0 * fabs(beta), sgn = beta < 0.0 ? -1.0 : 1.0;
  const float* Qc = tb->Q[src];
  const double* Dc = tb->Dc[src];
  double dmax = 0.0, zmax = 0.0;
  for (int i = tid; i < m; i += 256) {
    const double di = Dc[r0 + i];
    const double zi = 0.70710678118654752440 *
                      (i < n1 ? (double)Qc[(int64_t)(r0 + n1 - 1) * ld + r0 + i]
                              : sgn * (double)Qc[(int64_t)(r0 + n1) * ld + r0 + i]);
    d[i] = di; z[i] = zi;
    dmax = fmax(dmax, fabs(di));
    zmax = fmax(zmax, fabs(zi));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
    zmax = fmax(zmax, __shfl_xor(zmax, off, 64));
  }
  if ((tid & 63) == 0) { sred[0][tid >> 6] = dmax; sred[1][tid >> 6] = zmax; }
  __syncthreads();
  dmax = fmax(fmax(sred[0][0], sred[0][1]), fmax(sred[0][2], sred[0][3]));
  zmax = fmax(fmax(sred[1][0], sred[1][1]), fmax(sred[1][2], sred[1][3]));
  // merged ascending order of the two sorted halves (ties: first half first)
  for (int i = tid; i < m; i += 256) {
    const double di = d[i];
    int lo, hi, pos;
    if (i < n1) {   // count of second-half entries < di
      lo = n1; hi = m;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (d[mid] < di) lo = mid + 1; else hi = mid; }
      pos = i + (lo - n1);
    } else {        // count of first-half entries <= di
      lo = 0; hi = n1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (d[mid] <= di) lo = mid + 1; else hi = mid; }
      pos = (i - n1) + lo;
    }
    perm[pos] = i;
  }
  __syncthreads();
  if (tid != 0) return;
  const double tol = 8.0 * (double)eps_defl * fmax(dmax, zmax);
  psdc::DcDeflateOut o;
  if (rho * zmax <= tol) {
    o.K = 0; o.nrot = 0; o.ndefl = m;
    for (int i = 0; i < m; ++i) { tb->dflv[r0 + i] = d[perm[i]]; tb->dflc[r0 + i] = perm[i]; }
  } else {
    o = psdc::dc_deflate(m, perm, d, z, rho, tol, tb->dl + r0, tb->w + r0, tb->col + r0,
                         tb->dflv + r0, tb->dflc + r0, tb->ra + r0, tb->rb + r0, tb->rc + r0,
                         tb->rs + r0);
  }
  TdNodeSt s;
  s.K = o.K; s.ndefl = o.ndefl; s.nrot = o.nrot; s.pad_ = 0; s.rho = rho;
  st[blockIdx.x] = s;
}

// secular equation: one thread per root; poles and weights of the node in LDS
__global__ __launch_bounds__(256) void td_dc_secular_kernel(TdBlock* blocks, const TdNode* nodes,
                                                           const TdNodeSt* st) {
  extern __shared__ __align__(16) unsigned char td_dyn[];
  const TdNode nd = nodes[blockIdx.x];
  const TdNodeSt s = st[blockIdx.x];
  const int K = s.K, r0 = nd.r0, tid = threadIdx.x;
  if ((int)blockIdx.y * 256 >= K) return;
  TdBlock* tb = &blocks[nd.blk];
  double* dl = reinterpret_cast<double*>(td_dyn);
  double* w = dl + K;
  for (int i = tid; i < K; i += 256) { dl[i] = tb->dl[r0 + i]; w[i] = tb->w[r0 + i]; }
  __syncthreads();
  const int j = blockIdx.y * 256 + tid;
  if (j >= K) return;
  int org; double mu;
  const int it = psdc::dc_secular_root(K, j, dl, w, s.rho, &org, &mu);
  tb->org[r0 + j] = org;
  tb->mu[r0 + j] = mu;
  if (it >= psdc::DC_SEC_MAXIT || !(mu == mu)) tb->fail = 1;
}

// Loewner weights: one thread per pole
__global__ __launch_bounds__(256) void td_dc_zhat_kernel(TdBlock* blocks, const TdNode* nodes,
                                                        const TdNodeSt* st) {
  extern __shared__ __align__(16) unsigned char td_dyn[];
  const TdNode nd = nodes[blockIdx.x];
  const TdNodeSt s = st[blockIdx.x];
  const int K = s.K, r0 = nd.r0, tid = threadIdx.x;
  if ((int)blockIdx.y * 256 >= K) return;
  TdBlock* tb = &blocks[nd.blk];
  double* dl = reinterpret_cast<double*>(td_dyn);
  double* w = dl + K;
  double* mu = w + K;
  int* org = reinterpret_cast<int*>(mu + K);
  for (int i = tid; i < K; i += 256) {
    dl[i] = tb->dl[r0 + i]; w[i] = tb->w[r0 + i]; mu[i] = tb->mu[r0 + i]; org[i] = tb->org[r0 + i];
  }
  __syncthreads();
  const int i = blockIdx.y * 256 + tid;
  if (i >= K) return;
  tb->zh[r0 + i] = psdc::dc_zhat(K, i, dl, w, org, mu);
}

// order of the new eigenvalues (roots and deflated values merged), maps for the S builder, 1 / |x_j|
__global__ __launch_bounds__(256) void td_dc_order_kernel(TdBlock* blocks, const TdNode* nodes,
                                                         const TdNodeSt* st, int dst) {
  extern __shared__ __align__(16) unsigned char td_dyn[];
  const TdNode nd = nodes[blockIdx.x];
  const TdNodeSt s = st[blockIdx.x];
  const int K = s.K, nd_ = s.ndefl, r0 = nd.r0, m = nd.m, tid = threadIdx.x;
  TdBlock* tb = &blocks[nd.blk];
  double* lam = reinterpret_cast<double*>(td_dyn);   // [K]
  double* dv = lam + K;                              // [ndefl]
  double* dl = dv + nd_;                             // [K]
  double* zh = dl + K;                               // [K]
  for (int i = tid; i < K; i += 256) {
    const double dli = tb->dl[r0 + i];
    dl[i] = dli; zh[i] = tb->zh[r0 + i];
  }
  for (int t = tid; t < nd_; t += 256) dv[t] = tb->dflv[r0 + t];
  __syncthreads();
  for (int i = tid; i < K; i += 256) lam[i] = dl[tb->org[r0 + i]] + tb->mu[r0 + i];
  __syncthreads();
  for (int e = blockIdx.y * 256 + tid; e < m; e += gridDim.y * 256) {
    int pos;
    double val;
    if (e < K) {
      val = lam[e];
      pos = e;
      for (int t = 0; t < nd_; ++t) pos += dv[t] < val ? 1 : 0;
      tb->rowmap[r0 + tb->col[r0 + e]] = e;
      tb->rn[r0 + e] = psdc::dc_vec_rnorm(K, e, dl, zh, tb->org[r0 + e], tb->mu[r0 + e]);
    } else {
      const int t = e - K;
      val = dv[t];
      pos = 0;
      for (int u = 0; u < nd_; ++u) pos += (dv[u] < val || (dv[u] == val && u < t)) ? 1 : 0;
      for (int a = 0; a < K; ++a) pos += lam[a] <= val ? 1 : 0;
      tb->rowmap[r0 + tb->dflc[r0 + t]] = e;
    }
    tb->Dc[dst][r0 + pos] = val;
    tb->colkind[r0 + pos] = e;
  }
}

// S[rho][kappa] of the node: zhat_i / (dl_i - lambda_j) / |x_j| on (kept row, root column), 1 on a
// deflated pair, 0 elsewhere.  grid (node, column chunk of 256, row chunk of 64)
__global__ __launch_bounds__(256) void td_dc_sbuild_kernel(TdBlock* blocks, const TdNode* nodes,
                                                          const TdNodeSt* st) {
  const TdNode nd = nodes[blockIdx.x];
  const int K = st[blockIdx.x].K, r0 = nd.r0, m = nd.m;
  const int kap = blockIdx.y * 256 + threadIdx.x;
  const int row_lo = blockIdx.z * 64;
  if ((int)blockIdx.y * 256 >= m || row_lo >= m) return;
  TdBlock* tb = &blocks[nd.blk];
  const int ld = tb->ld;
  const bool live = kap < m;
  const int ck = live ? tb->colkind[r0 + kap] : 0;
  double lam_o = 0.0, mu = 0.0, rn = 0.0;
  if (live && ck < K) {
    lam_o = tb->dl[r0 + tb->org[r0 + ck]];
    mu = tb->mu[r0 + ck];
    rn = tb->rn[r0 + ck];
  }
  const int row_hi = min(row_lo + 64, m);
  for (int rho = row_lo; rho < row_hi; ++rho) {
    const int rk = tb->rowmap[r0 + rho];   // uniform
    float val = 0.f;
    if (rk < K) {
      if (ck < K) {
        double dlt = (tb->dl[r0 + rk] - lam_o) - mu;
        if (dlt == 0.0) dlt = 1e-300;
        val = (float)(tb->zh[r0 + rk] / dlt * rn);
      }
    } else if (rk == ck) {
      val = 1.f;
    }
    if (live) tb->S[(int64_t)(r0 + rho) * ld + r0 + kap] = val;
  }
}

// plane rotations of the deflation, applied to the rows of S in reverse order
__global__ __launch_bounds__(256) void td_dc_rot_kernel(TdBlock* blocks, const TdNode* nodes,
                                                       const TdNodeSt* st) {
  const TdNode nd = nodes[blockIdx.x];
  const int nrot = st[blockIdx.x].nrot, r0 = nd.r0, m = nd.m;
  const int kap = blockIdx.y * 256 + threadIdx.x;
  if (nrot == 0 || kap >= m) return;
  TdBlock* tb = &blocks[nd.blk];
  const int ld = tb->ld;
  float* Sc = tb->S + (int64_t)r0 * ld + r0 + kap;
  for (int t = nrot - 1; t >= 0; --t) {
    const int a = tb->ra[r0 + t], b = tb->rb[r0 + t];
    const double c = tb->rc[r0 + t], s = tb->rs[r0 + t];
    const double xa = Sc[(int64_t)a * ld], xb = Sc[(int64_t)b * ld];
    Sc[(int64_t)a * ld] = (float)(c * xa - s * xb);
    Sc[(int64_t)b * ld] = (float)(s * xa + c * xb);
  }
}

// merge products: Qnext[rows of a half, columns of the node] = Q_half S[rows of the half, :]
template <bool GUARD>
__global__ __launch_bounds__(256, 2) void td_dc_gemm_kernel(TdBlock* blocks, const TdGTile* tiles,
                                                            int src) {
  __shared__ __align__(16) float smem[SmemCfg<16>::TOTAL];
  const TdGTile t = tiles[blockIdx.x];
  const TdBlock tbv = blocks[t.blk];   // by value: one burst of scalar loads
  const TdBlock* tb = &tbv;
  const int ld = tb->ld;
  const float* Qc = tb->Q[src];
  float* Qn = tb->Q[src ^ 1];
  const int row0 = t.r0h + t.tm * TILE, col0 = t.r0 + t.tn * TILE;
  Operand A{Qc + t.r0h, ld, row0, t.r0h + t.nh, t.nh, (t.r0h & 3) == 0};
  Operand B{tb->S + (int64_t)t.r0h * ld, ld, col0, t.r0 + t.m, t.nh, (col0 & 3) == 0};
  f32x16 acc[2][2];
  gemm_tile<KC, MC, 16, GUARD>(A, B, t.nh, smem, acc);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + acc_row(wm, a, r, lane), col = col0 + acc_col(wn, b, lane);
        if (GUARD && (row >= t.r0h + t.nh || col >= t.r0 + t.m)) continue;
        gstore1(Qn + (int64_t)row * ld + col, acc[a][b][r]);
      }
}

// eigenvalues back to the caller's scale; stage 1 (developer): Z_T = I, eigenvalues = diag(T)
// max_cond > 0 (root mode): the block's result stands only if lambda_max / lambda_min <= max_cond
// (the eigenvalues are ascending; a non-positive lambda_min fails the test).
__global__ __launch_bounds__(256) void td_dc_finish_kernel(TdBlock* blocks, int identity,
                                                          float max_cond) {
  TdBlock* tb = &blocks[blockIdx.y];
  const int n = tb->n, ld = tb->ld;
  const int src = tb->height & 1;
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < ld; i += 256)
      tb->evals[i] = i < n ? (identity ? tb->dT[i] : (float)(tb->Dc[src][i] * tb->scale)) : 0.f;
    if (threadIdx.x == 0) {
      int keep = tb->fail ? 0 : 1;
      if (keep && !identity && max_cond > 0.f) {
        const double lo = tb->Dc[src][0], hi = tb->Dc[src][n - 1];
        if (!(lo > 0.0) || !(hi <= (double)max_cond * lo)) keep = 0;
      }
      tb->keep = keep;
    }
  }
  if (!identity) return;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)n * n; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e % n);
    tb->Z[(int64_t)r * ld + c] = r == c ? 1.f : 0.f;
  }
}

// ======================= back-transformation Z <- Q Z ================================================

// T^-1 = diag(1 / tau) + striu(Vb Vb^T) of WY block kb (reflectors 128 kb ..): grid (4 nbk_max, nblk), one
// 64 x 64 quarter per workgroup (the quarter below the diagonal is not needed), one wavefront per 32 x 32
// quadrant on the float64 MFMA (products of float32 numbers are exact in float64).  Q = I - V^T T V is
// orthogonal exactly when T^-1 + T^-T = V V^T: a float32 Gram matrix leaves eps32 sqrt(n) |v_i . v_j|
// in that identity, which T amplifies by ||T||^2 -- 1e-6 of lost orthogonality on random inputs (nearly
// orthogonal reflectors, T ~ diag(tau)), 2e-5 on an all-ones matrix (consecutive reflectors nearly
// parallel); float64 accumulation brings both to the 1e-7 of the reflectors themselves at the same cost
// (0.5 ms of 200 at 64 x 2048^2).
__global__ __launch_bounds__(256) void td_vtv_kernel(TdBlock* blocks) {
  __shared__ float sL[RK][RQ + 1];
  __shared__ float sR[RK][RQ + 1];
  const TdBlock tbv = blocks[blockIdx.y];   // by value: one burst of scalar loads
  const TdBlock* tb = &tbv;
  const int n = tb->n, ld = tb->ld, kb = blockIdx.x >> 2, sub = blockIdx.x & 3, j0 = kb * TD_KB;
  if (j0 > n - 3 || !tb->keep || sub == 2) return;
  const int r0 = (sub >> 1) * RQ, c0 = (sub & 1) * RQ;   // quarter (rows r0 .., columns c0 ..) of the 128 x 128 block
  const float* Vb = tb->VHt + (int64_t)j0 * ld + j0;     // columns < j0 are zero
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int qr = 32 * (wave >> 1), qc = 32 * (wave & 1);
  const int fi = lane & 15, fk = lane >> 4;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int m = tid >> 2, k4 = (tid & 3) * 4;
  const int klen = ld - j0;   // a multiple of 128
  // reflector j0 + i is zero up to column j0 + i: rows r0 .. of the left operand start at column r0 + 1,
  // and the quarter needs only k >= max(r0, c0) (rounded down to the chunk)
  for (int k0 = (r0 > c0 ? r0 : c0); k0 < klen; k0 += RK) {
    const f32x4 vl = gload4(Vb + (int64_t)(r0 + m) * ld + k0 + k4);
    const f32x4 vr = gload4(Vb + (int64_t)(c0 + m) * ld + k0 + k4);
    sL[k4 + 0][m] = vl[0]; sL[k4 + 1][m] = vl[1]; sL[k4 + 2][m] = vl[2]; sL[k4 + 3][m] = vl[3];
    sR[k4 + 0][m] = vr[0]; sR[k4 + 1][m] = vr[1]; sR[k4 + 2][m] = vr[2]; sR[k4 + 3][m] = vr[3];
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < RK; kk += 4) {
      double af[2], bf[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = (double)sL[kk + fk][qr + 16 * a + fi];
#pragma unroll
      for (int b = 0; b < 2; ++b) bf[b] = (double)sR[kk + fk][qc + 16 * b + fi];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }
  float* T = tb->TT + (int64_t)kb * TD_KB * TD_KB;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = r0 + qr + 16 * a + fk + 4 * v, col = c0 + qc + 16 * b + fi;
        float x = 0.f;
        if (col > row) x = (float)acc[a][b][v];
        else if (col == row) {
          const int j = j0 + row;
          const float tau = j <= n - 2 ? tb->tau[j] : 0.f;
          x = tau != 0.f ? 1.f / tau : 1.f;
        }
        T[row * TD_KB + col] = x;
      }
}

// T = (T^-1)^-1, upper triangular, in LDS (float64 accumulation): grid (nbk_max, nblk), 128 threads
__global__ __launch_bounds__(128) void td_tinv_kernel(TdBlock* blocks) {
  extern __shared__ __align__(16) unsigned char td_dyn[];
  constexpr int LDM = TD_KB + 1;
  float* M = reinterpret_cast<float*>(td_dyn);
  float* X = M + TD_KB * LDM;
  const TdBlock tbv = blocks[blockIdx.y];   // by value: one burst of scalar loads
  const TdBlock* tb = &tbv;
  const int n = tb->n, kb = blockIdx.x, j0 = kb * TD_KB, k = threadIdx.x;
  if (j0 > n - 3 || !tb->keep) return;
  float* T = tb->TT + (int64_t)kb * TD_KB * TD_KB;
  for (int e = k; e < TD_KB * TD_KB; e += 128) {
    M[(e >> 7) * LDM + (e & 127)] = T[e];
    X[(e >> 7) * LDM + (e & 127)] = 0.f;
  }
  __syncthreads();
  // column k of the inverse by back substitution
  X[k * LDM + k] = 1.f / M[k * LDM + k];
  for (int r = k - 1; r >= 0; --r) {
    double s = 0.0;
    for (int q = r + 1; q <= k; ++q) s += (double)M[r * LDM + q] * (double)X[q * LDM + k];
    X[r * LDM + k] = (float)(-s / (double)M[r * LDM + r]);
  }
  __syncthreads();
  for (int e = k; e < TD_KB * TD_KB; e += 128) T[e] = X[(e >> 7) * LDM + (e & 127)];
}

// V2[j0 + i][c] = sum_i' T[i][i'] V[j0 + i'][c]: grid (nt_max, nbk_max, nblk)
__global__ __launch_bounds__(256, 2) void td_v2_kernel(TdBlock* blocks) {
  __shared__ __align__(16) float smem[SmemCfg<16>::TOTAL];
  const TdBlock tbv = blocks[blockIdx.z];   // by value: one burst of scalar loads
  const TdBlock* tb = &tbv;
  const int n = tb->n, ld = tb->ld, kb = blockIdx.y, j0 = kb * TD_KB, ct = blockIdx.x;
  if (j0 > n - 3 || !tb->keep || ct >= tb->nt || (ct + 1) * TILE <= j0) return;
  const float* T = tb->TT + (int64_t)kb * TD_KB * TD_KB;
  Operand A{T, TD_KB, 0, TD_KB, TD_KB, true};
  Operand B{tb->VHt + (int64_t)j0 * ld, ld, ct * TILE, ld, TD_KB, true};
  f32x16 acc[2][2];
  gemm_tile<KC, MC, 16, false>(A, B, TD_KB, smem, acc);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1;
  float* V2 = tb->S;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = j0 + acc_row(wm, a, r, lane), col = ct * TILE + acc_col(wn, b, lane);
        if (row < ld) gstore1(V2 + (int64_t)row * ld + col, acc[a][b][r]);
      }
}

// Y2 = V2_b Z (128 x ld): grid (nt_max, nblk)
__global__ __launch_bounds__(256, 2) void td_bt1_kernel(TdBlock* blocks, int kb) {
  __shared__ __align__(16) float smem[SmemCfg<16>::TOTAL];
  const TdBlock tbv = blocks[blockIdx.y];   // by value: one burst of scalar loads
  const TdBlock* tb = &tbv;
  const int n = tb->n, ld = tb->ld, j0 = kb * TD_KB, ct = blockIdx.x;
  if (j0 > n - 3 || !tb->keep || ct >= tb->nt) return;
  Operand A{tb->S + (int64_t)j0 * ld + j0, ld, 0, TD_KB, ld - j0, true};
  Operand B{tb->Z + (int64_t)j0 * ld, ld, ct * TILE, ld, ld - j0, true};
  f32x16 acc[2][2];
  gemm_tile<KC, MC, 16, false>(A, B, ld - j0, smem, acc);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = acc_row(wm, a, r, lane), col = ct * TILE + acc_col(wn, b, lane);
        gstore1(tb->Y2 + (int64_t)row * ld + col, acc[a][b][r]);
      }
}

// Z[c][col] -= sum_i V[j0 + i][c] Y2[i][col]: grid (nt_max (row tiles from j0 / 128) * nt_max, nblk)
__global__ __launch_bounds__(256, 2) void td_bt3_kernel(TdBlock* blocks, int kb, int ntmax) {
  __shared__ __align__(16) float smem[SmemCfg<16>::TOTAL];
  const TdBlock tbv = blocks[blockIdx.y];   // by value: one burst of scalar loads
  const TdBlock* tb = &tbv;
  const int n = tb->n, ld = tb->ld, j0 = kb * TD_KB;
  const int rt = kb + blockIdx.x / ntmax, ct = blockIdx.x % ntmax;
  if (j0 > n - 3 || !tb->keep || rt >= tb->nt || ct >= tb->nt) return;
  Operand A{tb->VHt + (int64_t)j0 * ld, ld, rt * TILE, ld, TD_KB, true};
  Operand B{tb->Y2, ld, ct * TILE, ld, TD_KB, true};
  f32x16 acc[2][2];
  gemm_tile<MC, MC, 16, false>(A, B, TD_KB, smem, acc);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * TILE + acc_row(wm, a, r, lane), col = ct * TILE + acc_col(wn, b, lane);
        float* dst = tb->Z + (int64_t)row * ld + col;
        gstore1(dst, gload1(dst) - acc[a][b][r]);
      }
}

__global__ void td_status_kernel(const int* nfail, EStatus* status) {
  status->active = nfail[0];   // blocks handed back to the Jacobi solvers
  status->pad_ = nfail[1];     // ... of which for an iteration cap / non-finite input
  status->max_off = 0.f;
  __threadfence_system();
  status->gen = 0;
}

// hand the result to the common finish of eigh.hip: V = Z (identity on the padding), A = diag(evals)
__global__ __launch_bounds__(256) void td_finalize_kernel(TdBlock* blocks, EighBlock* ebs,
                                                         int* nfail) {
  TdBlock* tb = &blocks[blockIdx.y];
  EighBlock* eb = &ebs[tb->eb];
  const int n = tb->n, ld = tb->ld, keep = tb->keep;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    eb->td_done = keep;
    if (keep) eb->active = 0;
    else {
      atomicAdd(&nfail[0], 1);
      if (tb->fail) atomicAdd(&nfail[1], 1);
    }
  }
  if (!keep) return;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)ld * ld; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / ld), c = (int)(e % ld);
    eb->A[e] = r == c ? tb->evals[r] : 0.f;
    if (r >= n || c >= n) eb->V[e] = r == c ? 1.f : 0.f;
    if (c == 0) eb->evals[r] = tb->evals[r];
  }
}

// before the common finish: the blocks solved here take the finish of the one-sided Jacobi path
__global__ void td_mark_cj_kernel(EighBlock* blocks, int nblocks) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  if (blocks[b].td_done) { blocks[b].cj = 1; blocks[b].cj_active = 0; blocks[b].active = 0; }
}

}  // namespace psk

// ======================= host side =====================================================================
namespace psk {

struct TdPlan {
  std::vector<int> ids;            // EighBlock index of every block solved here
  std::vector<int> n, ld, height;
  int nmax = 0, ntmax = 0, nbkmax = 0, hmax = 0, mmax = 0;
  std::vector<TdNode> leaves, cuts;
  std::vector<std::vector<TdNode>> lvl_nodes;     // [h - 1]
  std::vector<std::vector<TdGTile>> lvl_tiles;
  std::vector<int> lvl_mmax, lvl_unguarded;
  bool empty() const { return ids.empty(); }
};

inline void td_make_plan(TdPlan& pl, const std::vector<int>& ids, const std::vector<int>& n_eff,
                         const std::vector<int>& npad) {
  pl.ids = ids;
  for (size_t k = 0; k < ids.size(); ++k) {
    const int b = ids[k], n = n_eff[b];
    pl.n.push_back(n);
    pl.ld.push_back(npad[b]);
    std::vector<psdc::DcNode> nodes;
    int h = 0;
    psdc::dc_make_plan(n, nodes, &h);
    pl.height.push_back(h);
    pl.nmax = std::max(pl.nmax, n);
    pl.ntmax = std::max(pl.ntmax, npad[b] / TILE);
    pl.nbkmax = std::max(pl.nbkmax, (std::max(n - 2, 1) + TD_KB - 1) / TD_KB);
    pl.hmax = std::max(pl.hmax, h);
    if ((int)pl.lvl_nodes.size() < h) {
      pl.lvl_nodes.resize(h); pl.lvl_tiles.resize(h);
      pl.lvl_mmax.resize(h, 0); pl.lvl_unguarded.resize(h, 1);
    }
    for (const psdc::DcNode& nd : nodes) {
      if (nd.n1 == 0) { pl.leaves.push_back({(int)k, nd.r0, nd.m, 0}); continue; }
      const TdNode tn{(int)k, nd.r0, nd.m, nd.n1};
      pl.cuts.push_back(tn);
      const int lv = nd.height - 1;
      pl.lvl_nodes[lv].push_back(tn);
      pl.lvl_mmax[lv] = std::max(pl.lvl_mmax[lv], nd.m);
      pl.mmax = std::max(pl.mmax, nd.m);
      if ((nd.r0 % TILE) || (nd.n1 % TILE) || (nd.m % TILE)) pl.lvl_unguarded[lv] = 0;
      for (int half = 0; half < 2; ++half) {
        const int r0h = half ? nd.r0 + nd.n1 : nd.r0, nh = half ? nd.m - nd.n1 : nd.n1;
        for (int tm = 0; tm < (nh + TILE - 1) / TILE; ++tm)
          for (int tn2 = 0; tn2 < (nd.m + TILE - 1) / TILE; ++tn2)
            pl.lvl_tiles[lv].push_back({(int)k, nd.r0, nd.m, r0h, nh, (short)tm, (short)tn2});
      }
    }
  }
}

struct TdLayout {
  TdBlock* blocks = nullptr;
  TdNode *leaves = nullptr, *cuts = nullptr;
  std::vector<TdNode*> lvl_nodes;
  std::vector<TdNodeSt*> lvl_st;
  std::vector<TdGTile*> lvl_tiles;
  int* nfail = nullptr;
  std::vector<TdBlock> host;    // filled by the carve when it runs on real memory
};

// Workspace of the tridiagonalisation path (on top of the five matrices per block of eigh.hip).
inline void td_carve(const TdPlan& pl, psh::Arena& ar, TdLayout* lo) {
  if (pl.empty()) return;
  const size_t B = pl.ids.size();
  TdBlock* blocks = ar.take<TdBlock>(B);
  TdNode* leaves = ar.take<TdNode>(std::max<size_t>(pl.leaves.size(), 1));
  TdNode* cuts = ar.take<TdNode>(std::max<size_t>(pl.cuts.size(), 1));
  int* nfail = ar.take<int>(4);
  if (lo) { lo->blocks = blocks; lo->leaves = leaves; lo->cuts = cuts; lo->nfail = nfail; lo->host.resize(B); }
  for (int h = 0; h < pl.hmax; ++h) {
    TdNode* nd = ar.take<TdNode>(std::max<size_t>(pl.lvl_nodes[h].size(), 1));
    TdNodeSt* st = ar.take<TdNodeSt>(std::max<size_t>(pl.lvl_nodes[h].size(), 1));
    TdGTile* tl = ar.take<TdGTile>(std::max<size_t>(pl.lvl_tiles[h].size(), 1));
    if (lo) { lo->lvl_nodes.push_back(nd); lo->lvl_st.push_back(st); lo->lvl_tiles.push_back(tl); }
  }
  for (size_t k = 0; k < B; ++k) {
    const size_t ld = pl.ld[k], nt = ld / TILE;
    TdBlock tb;
    memset(&tb, 0, sizeof(tb));
    tb.Wt = ar.take<float>(TD_NB * ld);
    tb.ubuf = ar.take<float>(ld);
    tb.wp = ar.take<float>(ld);
    tb.slab = ar.take<float>(nt * nt * TILE);
    tb.part_ss = ar.take<double>(nt);
    tb.part_dot = ar.take<float>(nt);
    tb.part_ab = ar.take<float>(2 * TD_NB * 32);
    tb.dT = ar.take<float>(ld); tb.eT = ar.take<float>(ld); tb.tau = ar.take<float>(ld);
    tb.evals = ar.take<float>(ld);
    double** f64s[] = {&tb.d64, &tb.e64, &tb.Dc[0], &tb.Dc[1], &tb.z, &tb.dl, &tb.w, &tb.mu,
                       &tb.zh, &tb.rn, &tb.dflv, &tb.rc, &tb.rs};
    for (double** p : f64s) *p = ar.take<double>(ld);
    int** i32s[] = {&tb.perm, &tb.col, &tb.dflc, &tb.ra, &tb.rb, &tb.org, &tb.rowmap, &tb.colkind};
    for (int** p : i32s) *p = ar.take<int>(ld);
    tb.TT = ar.take<float>((size_t)((ld + TD_KB - 1) / TD_KB) * TD_KB * TD_KB);
    tb.Y2 = ar.take<float>((size_t)TD_KB * ld);
    if (lo) lo->host[k] = tb;
  }
}

// Enqueues the whole solver on `st` for the blocks of the plan.  stage 1 (developer): stop after
// the reduction (Z_T = I: the output vectors are Q, the values diag(T)).
inline int td_run(hipStream_t st, const TdPlan& pl, TdLayout& lo, EighBlock* d_ebs,
                  const std::vector<EighBlock>& hb, float eps_defl, int stage, float max_cond,
                  int stream_groups, int tail_cols) {
  const int B = (int)pl.ids.size();
  for (int k = 0; k < B; ++k) {
    TdBlock& tb = lo.host[k];
    const EighBlock& eb = hb[pl.ids[k]];
    tb.eb = pl.ids[k];
    tb.n = pl.n[k]; tb.ld = pl.ld[k]; tb.nt = pl.ld[k] / TILE; tb.height = pl.height[k];
    tb.A = eb.A; tb.VHt = eb.X; tb.S = eb.W;
    tb.Q[0] = (tb.height & 1) ? eb.A : eb.V;
    tb.Q[1] = (tb.height & 1) ? eb.V : eb.A;
    tb.Z = eb.V;
    tb.jtail = (tail_cols > 0) ? (tb.n <= tail_cols ? 0 : (tb.n - tail_cols + TD_NB - 1) / TD_NB * TD_NB) : tb.n;
  }
  PS_RC(psh::upload_async(st, lo.blocks, lo.host.data(), sizeof(TdBlock) * B));
  if (!pl.leaves.empty()) PS_RC(psh::upload_async(st, lo.leaves, pl.leaves.data(), sizeof(TdNode) * pl.leaves.size()));
  if (!pl.cuts.empty()) PS_RC(psh::upload_async(st, lo.cuts, pl.cuts.data(), sizeof(TdNode) * pl.cuts.size()));
  for (int h = 0; h < pl.hmax; ++h) {
    if (pl.lvl_nodes[h].empty()) continue;
    PS_RC(psh::upload_async(st, lo.lvl_nodes[h], pl.lvl_nodes[h].data(), sizeof(TdNode) * pl.lvl_nodes[h].size()));
    PS_RC(psh::upload_async(st, lo.lvl_tiles[h], pl.lvl_tiles[h].data(), sizeof(TdGTile) * pl.lvl_tiles[h].size()));
  }
  PS_HIP(hipMemsetAsync(lo.nfail, 0, 4 * sizeof(int), st));
  static std::once_flag attr_once;
  std::call_once(attr_once, [] {
    const int big = 144 * 1024;
    (void)hipFuncSetAttribute((const void*)td_dc_deflate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)td_dc_secular_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)td_dc_zhat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)td_dc_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)td_tinv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)td_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(((size_t)TD_TAIL * (TD_TAIL | 1) + 2 * TD_TAIL + 16) * sizeof(float)));
  });
  const dim3 b256(256), b128(128);
  const int nmax = pl.nmax, ntmax = pl.ntmax;
  // ---- reduction to tridiagonal form ----
  // The blocks are dealt to `stream_groups` (1 | 2) groups that reduce on two streams: a column costs
  // three dependent launches, of which only the mat-vec is HBM-bound; one group's short vector kernels
  // run beside the other group's mat-vec.
  constexpr int MAXG = 8;
  hipStream_t side[MAXG - 1] = {};   // from the library's shared pool (common.h psh::side_stream)
  static thread_local hipEvent_t ev_fork = nullptr, ev_join[MAXG - 1] = {};
  // measured (64 x 2048 / 64 x 1024 / 256 x 512, ms): 1 group 181 / 43.1 / 28.9, 2 groups 173.5 / 43.0 / 28.0,
  // 4 groups 170.3 / 47.8 / 28.0: short columns gain nothing from a third and fourth group
  // while an asynchronous collective is in flight (ps_collective_in_flight: the N > 1 exchange of the previous
  // phase on RCCL's own stream) at most two groups: the runtime has four hardware queues, and more live streams
  // than queues serialise groups that share one (common.h psh::side_stream)
  const int cap_groups = PiPlan::health().collectives.load() > 0 ? 2 : stream_groups;
  const int want = nmax >= 1536 ? cap_groups : std::min(cap_groups, 2);
  const int ngroups = std::max(1, std::min(std::min(want, MAXG), B));
  if (!ev_fork) PS_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  for (int g = 0; g + 1 < ngroups; ++g) {
    side[g] = psh::side_stream(g);
    if (!side[g]) return PS_EINTERNAL;
    if (!ev_join[g]) PS_HIP(hipEventCreateWithFlags(&ev_join[g], hipEventDisableTiming));
  }
  hipLaunchKernelGGL(td_zero_kernel, dim3(256, B), b256, 0, st, lo.blocks, 0);
  hipLaunchKernelGGL(td_zero_kernel, dim3(8, B), b256, 0, st, lo.blocks, 1);
  int gfirst[MAXG + 1];
  for (int g = 0; g <= ngroups; ++g) gfirst[g] = (int)((int64_t)B * g / ngroups);
  hipStream_t gs[MAXG];
  gs[0] = st;
  for (int g = 1; g < ngroups; ++g) gs[g] = side[g - 1];
  struct SideJoin {   // however this scope is left, the side streams are joined back into the caller's
    hipStream_t st; hipStream_t* side; hipEvent_t* ev; int n;
    ~SideJoin() {
      for (int g = 0; g < n; ++g)
        if (hipEventRecord(ev[g], side[g]) == hipSuccess) (void)hipStreamWaitEvent(st, ev[g], 0);
    }
  } join{st, side, ev_join, 0};
  if (ngroups > 1) {
    PS_HIP(hipEventRecord(ev_fork, st));
    for (int g = 1; g < ngroups; ++g) PS_HIP(hipStreamWaitEvent(side[g - 1], ev_fork, 0));
    join.n = ngroups - 1;
  }
  int jend = 0, mtail = 0;   // columns the streaming kernels reduce; largest tail
  for (int k = 0; k < B; ++k) {
    jend = std::max(jend, lo.host[k].jtail);
    mtail = std::max(mtail, lo.host[k].n - lo.host[k].jtail);
  }
  for (int j = 0; j < jend; ++j) {
    const int i = j % TD_NB;
    for (int g = 0; g < ngroups; ++g) {
      TdBlock* gb = lo.blocks + gfirst[g];
      const int Bg = gfirst[g + 1] - gfirst[g];
      hipLaunchKernelGGL(td_row_kernel, dim3(ntmax - j / TILE, Bg), b128, 0, gs[g], gb,
                         i > 0 ? j - 1 : -1, j);
      if (j <= nmax - 2) {
        const int T = ntmax - (j + 1) / TILE;
        hipLaunchKernelGGL(td_symv_kernel, dim3(T * (T + 1) / 2, Bg), b256, 0, gs[g], gb, j, T);
        hipLaunchKernelGGL(td_w_kernel, dim3(T, Bg), b128, 0, gs[g], gb, j);
      }
      if (i == TD_NB - 1 && j + 1 <= nmax - 1) {   // row j + 1 (even if it is the last: d[n - 1]) reads the updated matrix
        const int p = j - i, T = ntmax - (p + TD_NB) / TILE;
        hipLaunchKernelGGL(td_row_kernel, dim3(ntmax - (j + 1) / TILE, Bg), b128, 0, gs[g], gb, j, -1);
        hipLaunchKernelGGL(td_syr2k_kernel, dim3(T * (T + 1) / 2, Bg), b256, 0, gs[g], gb, p, T);
      }
    }
  }
  for (int g = 1; g < ngroups; ++g) {
    PS_HIP(hipEventRecord(ev_join[g - 1], side[g - 1]));
    PS_HIP(hipStreamWaitEvent(st, ev_join[g - 1], 0));
  }
  join.n = 0;
  if (mtail > 0)
    hipLaunchKernelGGL(td_tail_kernel, dim3(B), dim3(512), ((size_t)mtail * (mtail | 1) + 2 * mtail + 16) * sizeof(float),
                       st, lo.blocks);
  PS_LAUNCH_CHECK();
  // ---- divide and conquer ----
  if (stage == 1) {
    hipLaunchKernelGGL(td_dc_finish_kernel, dim3(64, B), b256, 0, st, lo.blocks, 1, 0.f);
  } else {
    hipLaunchKernelGGL(td_dc_init_kernel, dim3(B), b256, 0, st, lo.blocks);
    if (!pl.cuts.empty())
      hipLaunchKernelGGL(td_dc_cuts_kernel, dim3(((int)pl.cuts.size() + 255) / 256), b256, 0, st,
                         lo.blocks, lo.cuts, (int)pl.cuts.size());
    hipLaunchKernelGGL(td_dc_leaf_kernel, dim3(((int)pl.leaves.size() + 3) / 4), b256, 0, st,
                       lo.blocks, lo.leaves, (int)pl.leaves.size());
    for (int h = 1; h <= pl.hmax; ++h) {
      const int nn = (int)pl.lvl_nodes[h - 1].size(), mm = pl.lvl_mmax[h - 1];
      if (nn == 0) continue;
      const int src = (h - 1) & 1, chunks = (mm + 255) / 256;
      hipLaunchKernelGGL(td_dc_deflate_kernel, dim3(nn), b256, (size_t)mm * 20, st, lo.blocks,
                         lo.lvl_nodes[h - 1], lo.lvl_st[h - 1], src, eps_defl);
      hipLaunchKernelGGL(td_dc_secular_kernel, dim3(nn, chunks), b256, (size_t)mm * 16, st, lo.blocks,
                         lo.lvl_nodes[h - 1], lo.lvl_st[h - 1]);
      hipLaunchKernelGGL(td_dc_zhat_kernel, dim3(nn, chunks), b256, (size_t)mm * 28, st, lo.blocks,
                         lo.lvl_nodes[h - 1], lo.lvl_st[h - 1]);
      hipLaunchKernelGGL(td_dc_order_kernel, dim3(nn, chunks), b256, (size_t)mm * 32, st, lo.blocks,
                         lo.lvl_nodes[h - 1], lo.lvl_st[h - 1], h & 1);
      hipLaunchKernelGGL(td_dc_sbuild_kernel, dim3(nn, chunks, (mm + 63) / 64), b256, 0, st,
                         lo.blocks, lo.lvl_nodes[h - 1], lo.lvl_st[h - 1]);
      hipLaunchKernelGGL(td_dc_rot_kernel, dim3(nn, chunks), b256, 0, st, lo.blocks,
                         lo.lvl_nodes[h - 1], lo.lvl_st[h - 1]);
      const int ntile = (int)pl.lvl_tiles[h - 1].size();
      if (pl.lvl_unguarded[h - 1])
        hipLaunchKernelGGL(td_dc_gemm_kernel<false>, dim3(ntile), b256, 0, st, lo.blocks,
                           lo.lvl_tiles[h - 1], src);
      else
        hipLaunchKernelGGL(td_dc_gemm_kernel<true>, dim3(ntile), b256, 0, st, lo.blocks,
                           lo.lvl_tiles[h - 1], src);
    }
    hipLaunchKernelGGL(td_dc_finish_kernel, dim3(1, B), b256, 0, st, lo.blocks, 0, max_cond);
  }
  PS_LAUNCH_CHECK();
  // ---- back-transformation ----
  const int nbk = pl.nbkmax;
  hipLaunchKernelGGL(td_vtv_kernel, dim3(4 * nbk, B), b256, 0, st, lo.blocks);
  hipLaunchKernelGGL(td_tinv_kernel, dim3(nbk, B), b128, (size_t)2 * TD_KB * (TD_KB + 1) * sizeof(float),
                     st, lo.blocks);
  hipLaunchKernelGGL(td_v2_kernel, dim3(ntmax, nbk, B), b256, 0, st, lo.blocks);
  for (int kb = nbk - 1; kb >= 0 && stage != 2; --kb) {   // stage 2 (developer): Z = Z_T
    hipLaunchKernelGGL(td_bt1_kernel, dim3(ntmax, B), b256, 0, st, lo.blocks, kb);
    hipLaunchKernelGGL(td_bt3_kernel, dim3((ntmax - kb) * ntmax, B), b256, 0, st, lo.blocks, kb, ntmax);
  }
  hipLaunchKernelGGL(td_finalize_kernel, dim3(256, B), b256, 0, st, lo.blocks, d_ebs, lo.nfail);
  PS_LAUNCH_CHECK();
  return 0;
}

}  // namespace psk
