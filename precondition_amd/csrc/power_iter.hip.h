// power_iter.hip.h — batched power iteration (reference: power_iteration,
// DS:595-652): lambda_max estimate with the fixed seed vector, <= num_iters steps,
// stop when |s_new - s| <= tol.
//
// HBM-bound: a step has to read every matrix.  The matrices are symmetric
// (statistics and their regularised forms), so a step reads only the 128x128 tiles
// (I, J >= I) of the upper block triangle — (T+1)/(2T) of the bytes — and uses each
// off-diagonal tile twice out of LDS: u = A_IJ v_J goes to y_I and w = A_IJ^T v_I
// to y_J.  Two launches per step over the whole batch:
//   pi_mv_kernel  : one workgroup per tile, writes the partial products into a
//                   per-block slab P[X][Y][128] ("contribution of v_Y to y_X");
//   pi_red_kernel : one workgroup per block sums the slab in a fixed order,
//                   forms s = v.(A v) and ||A v||, takes the stop decision of
//                   DS:639 and writes the normalised iterate for the next step.
// No float atomics, no inter-workgroup traffic inside a launch: results are
// bit-reproducible.
//
// Symmetry is a per-call contract (ps_api.h PS_SYMMETRY_*), never assumed silently: the
// reference's power_iteration (DS:595) takes any square matrix, and in the int16 state
// mode the dequantized statistics are NOT symmetric (per-column scales).  sym_check_kernel
// compares a_ij with a_ji bit for bit (one pass over the matrices, before the first step)
// and sets a per-block flag; blocks that are not exactly symmetric read BOTH tiles (I,J)
// and (J,I) in their off-diagonal workgroups and form the full mat-vec.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "gemm_core.hip.h"

namespace psk {

constexpr int PT = 128;        // tile side
constexpr int PLD = PT + 1;    // LDS row stride (odd: column walks conflict-free)

struct PiBlock {
  const float* a;
  int lda;
  int n;          // effective size (rows/cols >= n are padding and ignored)
  int t;          // tiles per side = ceil(n / 128)
  int vec_ok;     // float4 row loads are legal
  float* vn;      // [t*128] normalised iterate (zero beyond n)
  float* P;       // [t][t][128] partial products of the current step
  float s_prev;   // s of the previous step (0 before the first)
  int stop_iter;  // -1 while running, else the step after which the loop ended
  float lambda;   // s_out
  int iters;      // steps executed
  const int* asym;  // *asym != 0: the block is not exactly symmetric (full mat-vec)
};

struct PiTile {
  int block;
  short I, J;  // J >= I
};

// ---- v0 -> normalised first iterate (DS:634 of step 0) --------------------------
static __global__ __launch_bounds__(256) void pi_init_kernel(PiBlock* blocks,
                                                            const float* v0) {
  __shared__ float red[4];
  PiBlock* pb = &blocks[blockIdx.x];
  const int n = pb->n, tid = threadIdx.x;
  float ss = 0.f;
  for (int j = tid; j < n; j += 256) ss += v0[j] * v0[j];
  ss = wave_sum_f32(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float nrm = sqrtf(((red[0] + red[1]) + red[2]) + red[3]);
  for (int j = tid; j < pb->t * PT; j += 256) pb->vn[j] = j < n ? v0[j] / nrm : 0.f;
}

// ---- exact symmetry test: a_ij == a_ji on every element of the unpadded part --------
// One workgroup per tile (I, J >= I); 64x64 sub-blocks, the partner block staged
// transposed through LDS so that both global reads are row-contiguous.
static __global__ __launch_bounds__(256) void sym_check_kernel(const PiBlock* blocks,
                                                              const PiTile* tiles,
                                                              int* asym) {
  __shared__ float tb[64][65];
  const PiTile te = tiles[blockIdx.x];
  const PiBlock* pb = &blocks[te.block];
  const int n = pb->n, lda = pb->lda, tid = threadIdx.x;
  const float* a = pb->a;
  int bad = 0;
  for (int sub = 0; sub < 4; ++sub) {
    const int r0 = te.I * PT + 64 * (sub >> 1), c0 = te.J * PT + 64 * (sub & 1);
    if (r0 >= n || c0 >= n || (te.I == te.J && r0 > c0)) continue;  // uniform
    for (int e = tid; e < 64 * 64; e += 256) {
      const int rr = e >> 6, cc = e & 63;
      const int gr = c0 + rr, gc = r0 + cc;
      tb[rr][cc] = (gr < n && gc < n) ? gload1(a + (int64_t)gr * lda + gc) : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      const int gr = r0 + r, gc = c0 + c;
      const float x = (gr < n && gc < n) ? gload1(a + (int64_t)gr * lda + gc) : 0.f;
      if (!(x == tb[c][r])) bad = 1;  // a NaN counts as asymmetric: full products, NaN out
    }
    __syncthreads();
  }
  if (bad) asym[te.block] = 1;  // racing writers all store 1
}

static __global__ void sym_fill_kernel(int* asym, int count, int value) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) asym[i] = value;
}

// Row dot products of one 128x128 tile with a 128-vector: u[r] = sum_c A[r0+r][c0+c] v[c],
// written to P_out[r].  Register streaming: lane = (row parity rh, float4 column group c4),
// a wavefront walks 32 rows two at a time with all 16 loads in flight.
__device__ inline void pi_tile_rowdot(const float* a, int lda, int n, int r0, int c0, bool fast,
                                      f32x4 vc, int wave, int c4, int rh, float* P_out) {
  f32x4 x[16];
  if (fast) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int r = 32 * wave + 2 * it + rh;
      x[it] = gload4(a + (int64_t)(r0 + r) * lda + c0 + 4 * c4);
    }
  } else {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int r = 32 * wave + 2 * it + rh;
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (r0 + r < n) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c0 + 4 * c4 + e < n) t[e] = gload1(a + (int64_t)(r0 + r) * lda + c0 + 4 * c4 + e);
      }
      x[it] = t;
    }
  }
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int r = 32 * wave + 2 * it + rh;
    float u = x[it][0] * vc[0];
    u += x[it][1] * vc[1];
    u += x[it][2] * vc[2];
    u += x[it][3] * vc[3];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) u += __shfl_xor(u, off, 64);  // within the half
    if (c4 == 0) P_out[r] = u;
  }
}

// ---- one tile of A v (and of A^T v for off-diagonal tiles) ------------------------
// Register streaming, no LDS staging of the tile: lane = (row parity, float4 column
// group); a wavefront walks 32 rows two at a time with all 16 loads in flight.
// u[r] (row dot products) are reduced across the 32 lanes of a half-wave; w[c]
// (column sums of A^T v_I) accumulate in 4 registers per lane over the rows the
// lane visits and are combined across wavefronts through 2 KB of LDS.
static __global__ __launch_bounds__(256) void pi_mv_kernel(PiBlock* blocks,
                                                          const PiTile* tiles) {
  __shared__ float vI[PT];
  __shared__ float wpart[4][PT];
  const PiTile te = tiles[blockIdx.x];
  PiBlock* pb = &blocks[te.block];
  if (pb->stop_iter >= 0) return;  // uniform: written only by earlier launches
  const int n = pb->n, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int c4 = lane & 31, rh = lane >> 5;
  const int r0 = te.I * PT, c0 = te.J * PT;
  const float* a = pb->a;
  const int lda = pb->lda;
  const bool offdiag = te.I != te.J;
  if (tid < PT) vI[tid] = pb->vn[r0 + tid];
  const f32x4 vj = *reinterpret_cast<const f32x4*>(pb->vn + c0 + 4 * c4);
  __syncthreads();
  const bool fast = pb->vec_ok && c0 + PT <= n && r0 + PT <= n;
  f32x4 x[16];
  if (fast) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int r = 32 * wave + 2 * it + rh;
      x[it] = gload4(a + (int64_t)(r0 + r) * lda + c0 + 4 * c4);
    }
  } else {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int r = 32 * wave + 2 * it + rh;
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (r0 + r < n) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c0 + 4 * c4 + e < n) t[e] = gload1(a + (int64_t)(r0 + r) * lda + c0 + 4 * c4 + e);
      }
      x[it] = t;
    }
  }
  float* P = pb->P;
  const int t = pb->t;
  f32x4 w = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int r = 32 * wave + 2 * it + rh;
    float u = x[it][0] * vj[0];
    u += x[it][1] * vj[1];
    u += x[it][2] * vj[2];
    u += x[it][3] * vj[3];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) u += __shfl_xor(u, off, 64);  // within the half
    if (c4 == 0) P[((int64_t)te.I * t + te.J) * PT + r] = u;
    const float vi = vI[r];
    w[0] += x[it][0] * vi; w[1] += x[it][1] * vi; w[2] += x[it][2] * vi; w[3] += x[it][3] * vi;
  }
  if (offdiag && *pb->asym != 0) {
    // not symmetric: the contribution of v_I to y_J needs the tile (J, I) itself
    const f32x4 vi4 = *reinterpret_cast<const f32x4*>(vI + 4 * c4);
    pi_tile_rowdot(a, lda, n, c0, r0, fast, vi4, wave, c4, rh,
                   P + ((int64_t)te.J * t + te.I) * PT);
  } else if (offdiag) {
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] += __shfl_xor(w[e], 32, 64);  // the two row parities
    if (rh == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) wpart[wave][4 * c4 + e] = w[e];
    }
    __syncthreads();
    if (tid < PT)
      P[((int64_t)te.J * t + te.I) * PT + tid] =
          ((wpart[0][tid] + wpart[1][tid]) + wpart[2][tid]) + wpart[3][tid];
  }
}

// ---- per block: y = sum of partials, s, ||y||, stop decision, next iterate ---------
static __global__ __launch_bounds__(512) void pi_red_kernel(PiBlock* blocks, int iter,
                                                           int num_iters, float tol) {
  extern __shared__ __align__(16) float ysm[];  // [t*128] y of this step
  __shared__ float red[16];
  PiBlock* pb = &blocks[blockIdx.x];
  if (pb->stop_iter >= 0) return;
  const int n = pb->n, t = pb->t, tid = threadIdx.x;
  if (n == 0) {  // all padding: v0 masked to zero, 0/0 (DS:634)
    if (tid == 0) {
      pb->lambda = __uint_as_float(0x7fc00000u);
      pb->iters = 1;
      pb->stop_iter = 0;
    }
    return;
  }
  float s = 0.f, ss = 0.f;
  for (int j = tid; j < t * PT; j += 512) {
    const int X = j >> 7, r = j & 127;
    const float* p = pb->P + (int64_t)X * t * PT + r;
    float y = 0.f;
    int Y = 0;
    for (; Y + 4 <= t; Y += 4) {  // independent loads, fixed summation order
      const float p0 = p[(Y + 0) * PT], p1 = p[(Y + 1) * PT], p2 = p[(Y + 2) * PT],
                  p3 = p[(Y + 3) * PT];
      y = (((y + p0) + p1) + p2) + p3;
    }
    for (; Y < t; ++Y) y += p[Y * PT];
    ysm[j] = y;
    s += pb->vn[j] * y;   // DS:637
    ss += y * y;
  }
  s = wave_sum_f32(s);
  ss = wave_sum_f32(ss);
  if ((tid & 63) == 0) { red[tid >> 6] = s; red[8 + (tid >> 6)] = ss; }
  __syncthreads();
  float s_new = 0.f, n2 = 0.f;
  for (int w = 0; w < 8; ++w) { s_new += red[w]; n2 += red[8 + w]; }
  const float nrm = sqrtf(n2);
  const bool run = fabsf(s_new - pb->s_prev) > tol;  // DS:639
  const bool last = iter + 1 >= num_iters;
  // next iterate, normalised (DS:634 of the next step / DS:651 at the end)
  for (int j = tid; j < t * PT; j += 512) pb->vn[j] = j < n ? ysm[j] / nrm : 0.f;
  __syncthreads();  // every thread has read s_prev before it is overwritten
  if (tid == 0) {
    pb->s_prev = s_new;
    if (!run || last) {
      pb->lambda = s_new;
      pb->iters = iter + 1;
      pb->stop_iter = iter;
    }
  }
}

static __global__ void pi_output_kernel(const PiBlock* blocks, int nblocks, float* out_lambda,
                                        int* out_iters, float* out_v, int ldv) {
  (void)nblocks;
  const int b = blockIdx.x;
  const PiBlock* pb = &blocks[b];
  if (threadIdx.x == 0) {
    if (out_lambda) out_lambda[b] = pb->lambda;
    if (out_iters) out_iters[b] = pb->iters;
  }
  if (out_v)
    for (int j = threadIdx.x; j < pb->n; j += blockDim.x) out_v[(int64_t)b * ldv + j] = pb->vn[j];
}

// ---- host side --------------------------------------------------------------------
struct PiPlan {
  int batch = 0, max_n = 0;
  std::vector<int> n_eff;
  std::vector<PiTile> tiles;
  PiBlock* d_blocks = nullptr;
  PiTile* d_tiles = nullptr;
  float* d_v0 = nullptr;
  int* d_asym = nullptr;   // [batch] 1 = not exactly symmetric
  std::vector<float*> d_vn, d_P;

  void build(int b, const std::vector<int>& ne) {
    batch = b;
    n_eff = ne;
    tiles.clear();
    max_n = 0;
    for (int i = 0; i < b; ++i) {
      max_n = std::max(max_n, ne[i]);
      const int t = (ne[i] + PT - 1) / PT;
      for (int I = 0; I < t; ++I)
        for (int J = I; J < t; ++J) tiles.push_back({i, (short)I, (short)J});
    }
  }

  void carve(psh::Arena& ar, bool assign) {
    PiBlock* blk = ar.take<PiBlock>(batch);
    PiTile* tl = ar.take<PiTile>(std::max<size_t>(tiles.size(), 1));
    float* v0 = ar.take<float>(std::max(max_n, 1));
    int* asym = ar.take<int>(std::max(batch, 1));
    if (assign) { d_blocks = blk; d_tiles = tl; d_v0 = v0; d_asym = asym; d_vn.clear(); d_P.clear(); }
    for (int i = 0; i < batch; ++i) {
      const int t = (n_eff[i] + PT - 1) / PT;
      float* vn = ar.take<float>(std::max(t * PT, 1));
      float* P = ar.take<float>(std::max(t * t * PT, 1));
      if (assign) { d_vn.push_back(vn); d_P.push_back(P); }
    }
  }

  // Uploads the descriptors through the pinned staging ring (no stream synchronisation).
  int upload(hipStream_t st, const float* const* a, const int32_t* lda) {
    std::vector<PiBlock> h(batch);
    for (int i = 0; i < batch; ++i) {
      PiBlock& pb = h[i];
      memset(&pb, 0, sizeof(pb));
      pb.a = a[i];
      pb.lda = lda[i];
      pb.n = n_eff[i];
      pb.t = (n_eff[i] + PT - 1) / PT;
      pb.vec_ok = (((uintptr_t)a[i] % 16 == 0) && (lda[i] % 4 == 0)) ? 1 : 0;
      pb.vn = d_vn[i];
      pb.P = d_P[i];
      pb.stop_iter = -1;
      pb.asym = d_asym + i;
    }
    std::vector<float> v0(std::max(max_n, 1));
    ps_power_iteration_v0(max_n, v0.data());
    PS_RC(psh::upload_async(st, d_blocks, h.data(), sizeof(PiBlock) * batch));
    PS_RC(psh::upload_async(st, d_v0, v0.data(), sizeof(float) * v0.size()));
    if (!tiles.empty())
      PS_RC(psh::upload_async(st, d_tiles, tiles.data(), sizeof(PiTile) * tiles.size()));
    return 0;
  }

  // Fills the per-block asymmetry flags according to the call's symmetry contract
  // (PS_SYMMETRY_VERIFY: test every block; _ASSUME: caller guarantees exact symmetry;
  // _GENERAL: full products everywhere).  Must follow upload().
  int enqueue_symmetry(hipStream_t st, int symmetry) {
    if (batch == 0) return 0;
    const int fill = symmetry == PS_SYMMETRY_GENERAL ? 1 : 0;
    hipLaunchKernelGGL(sym_fill_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, d_asym,
                       batch, fill);
    if (symmetry == PS_SYMMETRY_VERIFY && !tiles.empty())
      hipLaunchKernelGGL(sym_check_kernel, dim3((unsigned)tiles.size()), dim3(256), 0, st,
                         d_blocks, d_tiles, d_asym);
    PS_LAUNCH_CHECK();
    return 0;
  }

  // Enqueues the whole iteration: 2 launches per step, fixed count (data-dependent
  // stops are taken on the device; stopped blocks' workgroups exit at once).
  int enqueue(hipStream_t st, int num_iters, float tol) {
    const size_t shm = 0;
    const size_t red_shm = (size_t)((max_n + PT - 1) / PT) * PT * sizeof(float);  // <= 64 KB
    if (batch == 0) return 0;
    hipLaunchKernelGGL(pi_init_kernel, dim3(batch), dim3(256), 0, st, d_blocks, d_v0);
    const int nt = (int)tiles.size();
    for (int i = 0; i < num_iters; ++i) {
      if (nt > 0)
        hipLaunchKernelGGL(pi_mv_kernel, dim3(nt), dim3(256), shm, st, d_blocks, d_tiles);
      hipLaunchKernelGGL(pi_red_kernel, dim3(batch), dim3(512), red_shm, st, d_blocks, i,
                         num_iters, tol);
    }
    PS_LAUNCH_CHECK();
    return 0;
  }
};

}  // namespace psk
