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
#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"
#include "gemm_core.hip.h"
#include "options.h"

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
  int team;         // resident execution: workgroups in the block's team
  int expired;      // resident execution: a bounded wait ran out (result NaN, see PiHealth)
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
    if (pb->vec_ok && r0 + 64 <= n && c0 + 64 <= n) {
      // interior 64 x 64 pair: both blocks as float4 rows, all eight loads of a thread in flight (the scalar
      // loops below ran the check at 1.9 TB/s: 0.13 ms of a 14 ms recompute)
      const int rr = tid >> 4, c4 = (tid & 15) * 4;       // rows rr + 16 u, columns c4 .. c4 + 3
      f32x4 t[4], x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        t[u] = gload4(a + (int64_t)(c0 + rr + 16 * u) * lda + r0 + c4);   // block (J, I)
        x[u] = gload4(a + (int64_t)(r0 + rr + 16 * u) * lda + c0 + c4);   // block (I, J)
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) tb[rr + 16 * u][c4 + k] = t[u][k];
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (!(x[u][k] == tb[c4 + k][rr + 16 * u])) bad = 1;   // a NaN counts as asymmetric
      __syncthreads();
      continue;
    }
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

// ---- shared tile arithmetic (streaming and resident kernels: identical, term for term) ----
// Dot product of a lane's 4 matrix columns with its 4 vector elements (fused multiply-adds).
__device__ __forceinline__ float pi_dot4(const f32x4& x, const f32x4& v) {
  return __fmaf_rn(x[3], v[3], __fmaf_rn(x[2], v[2], __fmaf_rn(x[1], v[1], __fmul_rn(x[0], v[0]))));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float pi_dpp_add(float x) {
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROWMASK, 0xf, false);
  return x + __int_as_float(r);
}
// Sum over the 32 lanes of a half-wavefront with DPP adds only: rotations by 8, 4, 2, 1
// inside each row of 16 lanes (every lane of the row then holds the row's sum: each step
// adds a commuting pair), then lane 15 of the lower row broadcast into the upper row
// (row_bcast15).  The total is valid in the UPPER row of each half: lanes 16-31 / 48-63.
__device__ __forceinline__ float pi_half_sum(float x) {
  x = pi_dpp_add<0x128, 0xf>(x);  // row_ror:8
  x = pi_dpp_add<0x124, 0xf>(x);  // row_ror:4
  x = pi_dpp_add<0x122, 0xf>(x);  // row_ror:2
  x = pi_dpp_add<0x121, 0xf>(x);  // row_ror:1
  x = pi_dpp_add<0x142, 0xa>(x);  // row_bcast:15 into rows 1 and 3
  return x;
}
constexpr int PI_SUM_LANE = 16;   // (lane & 31) of the lanes that hold pi_half_sum's total

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
    const float u = pi_half_sum(pi_dot4(x[it], vc));
    if (c4 == PI_SUM_LANE) P_out[r] = u;
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
    const float u = pi_half_sum(pi_dot4(x[it], vj));
    if (c4 == PI_SUM_LANE) P[((int64_t)te.I * t + te.J) * PT + r] = u;
    const float vi = vI[r];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = __fmaf_rn(x[it][e], vi, w[e]);
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

// ==================================================================================
// Resident execution: the whole iteration in ONE launch, matrices held in registers.
// ==================================================================================
// The streaming execution above re-reads every matrix from HBM in each of the <= 100 steps
// (173 MB per step for 256 x 512^2: 29 us at the HBM ceiling) and pays two launches per step.
// But the upper block triangles of a whole batch fit the register files of the chip
// (256 CUs x 512 KB): here every workgroup keeps up to PI_RNT 128x128 tiles (64 VGPRs per
// tile) in registers for the entire iteration, and the workgroups of a block -- its "team" --
// exchange only 128-float vectors through the memory-side cache:
//   step:  tile mat-vecs from registers -> partial products P[X][Y]      (hop 1)
//          the owner of block row X (the workgroup holding tile (X,X)) sums its row in
//          pi_red_kernel's order and publishes y_X                        (hop 2)
//          every workgroup gathers y and redoes the block's scalar reduction (s = v.y,
//          ||y||, stop decision) and keeps the next iterate in LDS.
// A step costs two hand-off latencies instead of a pass over HBM.  The arithmetic is the
// streaming kernels' own (same per-tile code; the 512-thread reduction of pi_red_kernel is
// emulated by 256 threads x 2 virtual threads): results are bit-identical.
//
// Hand-off = 8-byte granules {value, tag} written and polled with agent-coherent (sc1)
// accesses (MI355X_MICROARCH.md, handoff-1to1): the tag is the step number + 1, the slabs are
// zeroed by the host before the launch, so a granule validates itself -- no counters, no
// cache-wide release / acquire fences (an agent-scope fence writes back / invalidates a whole
// L2 on this multi-XCD part: 40 us per step measured with the fence protocol, 19 us with
// counter + sc1 loads).  Slabs are double buffered by step parity: a workgroup can run at most
// one step ahead of its slowest team mate (it needs every team mate's step-k data to finish
// step k), so the slot of step k+2 is free when it is written.
// Teams spin, so they must be co-resident: the host launches at most the resident capacity
// of the chip per pass; blocks whose team exceeds it use the streaming execution.  Every
// spin is bounded by ONE deadline per launch (PS_PI_TIMEOUT_MS, default 5 s) and by the team's
// abort granule; on expiry the block's result is NaN, the event is counted in pinned host
// memory (PiPlan::health), the root drivers re-run the call on the streaming kernels when they
// see the count move at their first host wait, and the process stays on the streaming
// execution from then on.
typedef unsigned long long pi_granule;

// local: every workgroup of the team runs on ONE XCD (verified at run time, pi_team_is_local).  An
// `sc1` store drops the line from the XCD's L2 (MI355X_MICROARCH.md, "stores of each flavour"): a
// reader on the SAME XCD then fetches it at the cross-XCD rate.  A workgroup-scope store (`sc0`)
// keeps the line in that L2, where the team mates' `sc1` loads (which bypass only their L1) find it:
// the hand-off costs an L2 round trip instead of a fabric one.  Only valid when no reader sits
// on another XCD -- their L2 would never see the line.
__device__ __forceinline__ void pi_publish(pi_granule* p, float v, unsigned tag, bool local = false) {
  const pi_granule g = ((pi_granule)tag << 32) | (pi_granule)__float_as_uint(v);
  if (local) __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ pi_granule pi_peek(const pi_granule* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Everything a bounded wait needs: ONE deadline for the whole kernel (constant 100 MHz clock),
// the team's abort granule (the first member whose wait expires publishes it, the others poll
// it, so a team leaves together instead of each member spinning out its own bound), and the
// thread's own flag: once it is set every later wait returns at once.
struct PiWait {
  unsigned long long deadline;
  const pi_granule* abort;
  bool dead;
};
// Address = wave-uniform base + a 32-bit byte offset per lane: keeps the uniform part of the
// hand-off addresses in SGPRs (saddr form) instead of one hoisted 64-bit VGPR pair per granule
// stream -- with three tiles in registers those pairs were what spilled to scratch.
template <typename T>
__device__ __forceinline__ T* pi_at(T* ubase, uint32_t byte_off) {
  return (T*)((char*)ubase + (uint64_t)byte_off);
}
// Polls until the granule carries `tag`.
__device__ __forceinline__ float pi_await(const pi_granule* p, pi_granule g, unsigned tag,
                                          PiWait* w) {
  if (w->dead) return __uint_as_float((unsigned)g);
  int spins = 0;
  while ((unsigned)(g >> 32) != tag) {
    if ((++spins & 255) == 0) {
      if (__builtin_amdgcn_s_memrealtime() > w->deadline || pi_peek(w->abort) != 0) {
        w->dead = true;
        break;
      }
    }
    __builtin_amdgcn_s_sleep(1);
    g = pi_peek(p);
  }
  return __uint_as_float((unsigned)g);
}

#ifndef PI_RNT
#define PI_RNT 3
#endif
#ifndef PI_ONE_HOP_T
#define PI_ONE_HOP_T 4   // blocks of up to this many tiles per side exchange in one hop
#endif
struct PiTeamWG {
  int block;
  short ntile;   // tiles held by this workgroup (1..PI_RNT); 0: all-padding block
  short lead;    // 1: writes the block's results
  short I[4], J[4];
  short member, tsize;   // index of this workgroup in its team, workgroups of the team
};

__device__ inline void pi_load_tile(const PiBlock* pb, int r0, int c0, int wave, int c4, int rh,
                                    f32x4 (&x)[16]) {
  const int n = pb->n, lda = pb->lda;
  const float* a = pb->a;
  const bool fast = pb->vec_ok && c0 + PT <= n && r0 + PT <= n;
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
}

// One tile's contributions from registers: the arithmetic of pi_mv_kernel, term for term.
// P: granule slab of this step, [t][t][128].  The 128 row sums u (produced two lanes at a
// time) and the column sums are staged in LDS and published by 128 consecutive lanes, so a
// vector leaves the CU as whole 128-byte lines (scattered 16-byte sc1 stores made the steps
// transaction-bound).  stage: [2][128] floats of LDS owned by this tile slot.
__device__ inline void pi_tile_from_regs(const PiBlock* pb, const f32x4 (&x)[16], int I, int J,
                                         const float* vn, pi_granule* P, unsigned tag,
                                         float (*wpart)[PT], float (*stage)[PT], bool asym,
                                         int tid, bool local) {
  const int wave = tid >> 6, lane = tid & 63;
  const int c4 = lane & 31, rh = lane >> 5;
  const int r0 = I * PT, c0 = J * PT, t = pb->t;
  const bool offdiag = I != J;
  const float* vI = vn + r0;
  const f32x4 vj = *reinterpret_cast<const f32x4*>(vn + c0 + 4 * c4);
  f32x4 w = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int r = 32 * wave + 2 * it + rh;
    const float u = pi_half_sum(pi_dot4(x[it], vj));
    if (c4 == PI_SUM_LANE) stage[0][r] = u;
    const float vi = vI[r];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = __fmaf_rn(x[it][e], vi, w[e]);
  }
  if (offdiag && asym) {
    // the mirrored tile is streamed from memory, a few rows at a time (rare path: the tiles
    // held in registers leave no room for pi_tile_rowdot's 16 loads in flight); the per-row
    // arithmetic is pi_tile_rowdot's
    const f32x4 vi4 = *reinterpret_cast<const f32x4*>(vI + 4 * c4);
    const bool fast = pb->vec_ok && c0 + PT <= pb->n && r0 + PT <= pb->n;
    const float* a = pb->a;
    const int lda = pb->lda, n = pb->n;
#pragma unroll 2
    for (int it = 0; it < 16; ++it) {
      const int r = 32 * wave + 2 * it + rh;
      f32x4 xx = {0.f, 0.f, 0.f, 0.f};
      if (fast) {
        xx = gload4(a + (int64_t)(c0 + r) * lda + r0 + 4 * c4);
      } else if (c0 + r < n) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (r0 + 4 * c4 + e < n) xx[e] = gload1(a + (int64_t)(c0 + r) * lda + r0 + 4 * c4 + e);
      }
      const float u = pi_half_sum(pi_dot4(xx, vi4));
      if (c4 == PI_SUM_LANE) stage[1][r] = u;
    }
  } else if (offdiag) {
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] += __shfl_xor(w[e], 32, 64);
    if (rh == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) wpart[wave][4 * c4 + e] = w[e];
    }
  }
  __syncthreads();
  if (tid < PT) {
    pi_publish(pi_at(P + ((int64_t)I * t + J) * PT, 8u * (uint32_t)tid), stage[0][tid], tag, local);
    if (offdiag)
      pi_publish(pi_at(P + ((int64_t)J * t + I) * PT, 8u * (uint32_t)tid),
                 asym ? stage[1][tid]
                      : ((wpart[0][tid] + wpart[1][tid]) + wpart[2][tid]) + wpart[3][tid],
                 tag, local);
  }
}

template <int NT>
static __global__ __launch_bounds__(256, NT >= 3 ? 2 : (NT == 2 ? 3 : 4)) void pi_resident_kernel(
    PiBlock* blocks, const PiTeamWG* wgs, int num_iters, float tol,
    unsigned long long timeout_ticks, unsigned* expired_total) {
  extern __shared__ __align__(16) float pi_lds[];  // vn[tp] | y[tp], tp = t*128 of the largest block
  __shared__ float wpart[NT][4][PT];
  __shared__ float stage[NT][2][PT];
  __shared__ float red[16];
  __shared__ int s_dead;
  const PiTeamWG te = wgs[blockIdx.x];
  if (te.block < 0) return;
  PiBlock* pb = &blocks[te.block];
  const int n = pb->n, t = pb->t, tp = t * PT, tid = threadIdx.x;
  if (n == 0) {  // all padding: v0 masked to zero, 0/0 (DS:634)
    if (te.lead && tid == 0) {
      pb->lambda = __uint_as_float(0x7fc00000u);
      pb->iters = 1;
      pb->stop_iter = 0;
    }
    return;
  }
  float* vn = pi_lds;
  float* ysm = pi_lds + tp;
  for (int j = tid; j < tp; j += 256) vn[j] = pb->vn[j];
  if (tid == 0) s_dead = 0;
  const int wave = tid >> 6, lane = tid & 63;
  f32x4 x[NT][16];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    if (k < te.ntile) {
      pi_load_tile(pb, te.I[k] * PT, te.J[k] * PT, wave, lane & 31, lane >> 5, x[k]);
    } else {
#pragma unroll
      for (int it = 0; it < 16; ++it) x[k][it] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  const bool asym = *pb->asym != 0;
  pi_granule* Pbase = reinterpret_cast<pi_granule*>(pb->P);
  const int64_t slab = (int64_t)t * t * PT;            // P granules per parity
  pi_granule* Ybase = Pbase + 2 * slab;                // y granules: [2][tp]
  pi_granule* abort_g = Ybase + 2 * tp;                // one granule behind them (zeroed per call)
  pi_granule* place_g = abort_g + 1;                   // [tsize]: XCC id of every team member
  float s_prev = 0.f;
  PiWait wt{__builtin_amdgcn_s_memrealtime() + timeout_ticks, abort_g, timeout_ticks == 0};
  // Where did the dispatcher put this team?  The host lays teams out so that the workgroups of one
  // team are 8 launch indices apart (PiPlan::layout_for_xcds), which on this part puts them on one
  // XCD -- an observed property of the dispatcher, not a guarantee -- so it is verified: every
  // member publishes its XCC id (agent scope: valid for any placement), reads its team mates' and
  // the team hands off through the XCD's L2 only if all ids agree.  One ~3 us exchange per launch.
  __shared__ int s_local;
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu;   // HW_REG_XCC_ID
    pi_publish(place_g + te.member, __uint_as_float(xcc), 1u);
    int same = 1;
    for (int m = 0; m < te.tsize; ++m) {
      const float o = pi_await(place_g + m, pi_peek(place_g + m), 1u, &wt);
      same &= (__float_as_uint(o) == xcc) ? 1 : 0;
    }
    s_local = (same && !wt.dead && te.tsize > 1) ? 1 : 0;
  }
  __syncthreads();
  const bool local = s_local != 0;
  for (int iter = 0; iter < num_iters; ++iter) {
    const unsigned tag = (unsigned)iter + 1u;
    pi_granule* P = Pbase + (int64_t)(iter & 1) * slab;
    pi_granule* Yg = Ybase + (int64_t)(iter & 1) * tp;
    // ---- hop 1: partial products of this workgroup's tiles ----
#pragma unroll
    for (int k = 0; k < NT; ++k)
      if (k < te.ntile) {
        // the tile coordinates are re-read as opaque values in every step: everything derived
        // from them (LDS and slab addresses of three tiles) is then recomputed with a few
        // scalar / VALU adds per step instead of being hoisted out of the iteration loop into
        // VGPRs that the 192 registers of matrix data leave no room for (they went to scratch)
        int Ik = te.I[k], Jk = te.J[k];
        asm volatile("" : "+s"(Ik), "+s"(Jk));
        pi_tile_from_regs(pb, x[k], Ik, Jk, vn, P, tag, wpart[k], stage[k], asym, tid, local);
      }
    float sv[2] = {0.f, 0.f}, ssv[2] = {0.f, 0.f};
    if (t <= PI_ONE_HOP_T) {
      // small blocks: every workgroup sums the whole slab itself (one hand-off latency per
      // step; t*t*128 granules = 16 KB at t = 4)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        for (int j = tid + 256 * v; j < tp; j += 512) {
          const int X = j >> 7, r = j & 127;
          const pi_granule* p = P + (int64_t)X * t * PT + r;
          float y = 0.f;
          int Y = 0;
          for (; Y + 4 <= t; Y += 4) {  // independent loads, pi_red_kernel's summation order
            const pi_granule g0 = pi_peek(p + (Y + 0) * PT), g1 = pi_peek(p + (Y + 1) * PT),
                             g2 = pi_peek(p + (Y + 2) * PT), g3 = pi_peek(p + (Y + 3) * PT);
            const float p0 = pi_await(p + (Y + 0) * PT, g0, tag, &wt);
            const float p1 = pi_await(p + (Y + 1) * PT, g1, tag, &wt);
            const float p2 = pi_await(p + (Y + 2) * PT, g2, tag, &wt);
            const float p3 = pi_await(p + (Y + 3) * PT, g3, tag, &wt);
            y = (((y + p0) + p1) + p2) + p3;
          }
          for (; Y < t; ++Y) y += pi_await(p + Y * PT, pi_peek(p + Y * PT), tag, &wt);
          ysm[j] = y;
          sv[v] += vn[j] * y;   // DS:637
          ssv[v] += y * y;
        }
      }
    } else {
      // ---- hop 2: rows owned by this workgroup (it holds their diagonal tile) ----
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        if (k < te.ntile && te.I[k] == te.J[k] && tid < PT) {
          const int X = te.I[k];
          const pi_granule* p = pi_at(P + (int64_t)X * t * PT, 8u * (uint32_t)tid);
          float y = 0.f;
          int Y = 0;
          for (; Y + 4 <= t; Y += 4) {
            const pi_granule g0 = pi_peek(p + (Y + 0) * PT), g1 = pi_peek(p + (Y + 1) * PT),
                             g2 = pi_peek(p + (Y + 2) * PT), g3 = pi_peek(p + (Y + 3) * PT);
            const float p0 = pi_await(p + (Y + 0) * PT, g0, tag, &wt);
            const float p1 = pi_await(p + (Y + 1) * PT, g1, tag, &wt);
            const float p2 = pi_await(p + (Y + 2) * PT, g2, tag, &wt);
            const float p3 = pi_await(p + (Y + 3) * PT, g3, tag, &wt);
            y = (((y + p0) + p1) + p2) + p3;
          }
          for (; Y < t; ++Y) y += pi_await(p + Y * PT, pi_peek(p + Y * PT), tag, &wt);
          pi_publish(pi_at(Yg + X * PT, 8u * (uint32_t)tid), y, tag, local);
        }
      }
      // ---- gather y; pi_red_kernel's reduction, its 512 threads as 2 virtual threads each ----
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        for (int j = tid + 256 * v; j < tp; j += 512) {
          const pi_granule* yp = pi_at(Yg, 8u * (uint32_t)j);
          const float y = pi_await(yp, pi_peek(yp), tag, &wt);
          ysm[j] = y;
          sv[v] += vn[j] * y;   // DS:637
          ssv[v] += y * y;
        }
      }
    }
    if (wt.dead) {
      s_dead = 1;
      pi_publish(abort_g, 0.f, 1u);   // the whole team leaves with this step
    }
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const float a = wave_sum_f32(sv[v]), b = wave_sum_f32(ssv[v]);
      if (lane == 0) { red[4 * v + wave] = a; red[8 + 4 * v + wave] = b; }
    }
    __syncthreads();
    float s_new = 0.f, n2 = 0.f;
    for (int w = 0; w < 8; ++w) { s_new += red[w]; n2 += red[8 + w]; }
    const bool expired = s_dead != 0;
    if (expired) s_new = __uint_as_float(0x7fc00000u);
    const float nrm = sqrtf(n2);
    const bool run = fabsf(s_new - s_prev) > tol;  // DS:639 (false for NaN: the loop ends)
    const bool last = iter + 1 >= num_iters;
    for (int j = tid; j < tp; j += 256) vn[j] = j < n ? ysm[j] / nrm : 0.f;
    __syncthreads();
    s_prev = s_new;
    if (!run || last || expired) {
      if (te.lead) {
        for (int j = tid; j < tp; j += 256) pb->vn[j] = vn[j];
        if (tid == 0) {
          pb->s_prev = s_new;
          pb->lambda = s_new;
          pb->iters = iter + 1;
          pb->stop_iter = iter;
          if (expired) {   // host-visible (pinned) count: the drivers re-run on the streaming kernels
            pb->expired = 1;
            atomicAdd_system(expired_total, 1u);
            __threadfence_system();
          }
        }
      }
      break;
    }
  }
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
  // resident execution (pi_resident_kernel): team workgroups, split into co-resident passes
  static constexpr int RNT = PI_RNT;     // tiles per workgroup
  std::vector<PiTeamWG> wgs;
  std::vector<int> team;                 // workgroups per block
  PiTeamWG* d_wgs = nullptr;
  char* d_region = nullptr;              // per-block iterates + slabs (zeroed per resident call)
  size_t region_bytes = 0;

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
    // teams: consecutive tiles of a block, RNT per workgroup; an all-padding block gets one
    // workgroup that only writes its NaN result
    wgs.clear();
    team.assign(b, 0);
    size_t k = 0;
    for (int i = 0; i < b; ++i) {
      const int t = (ne[i] + PT - 1) / PT, nt = t * (t + 1) / 2;
      if (nt == 0) {
        wgs.push_back({i, 0, 1, {0, 0, 0, 0}, {0, 0, 0, 0}, 0, 1});
        team[i] = 1;
        continue;
      }
      const int tsz = (nt + RNT - 1) / RNT;
      for (int f = 0; f < nt; f += RNT) {
        PiTeamWG w{i, (short)std::min(RNT, nt - f), (short)(f == 0 ? 1 : 0), {0, 0, 0, 0}, {0, 0, 0, 0},
                   (short)(f / RNT), (short)tsz};
        for (int e = 0; e < w.ntile; ++e) { w.I[e] = tiles[k + f + e].I; w.J[e] = tiles[k + f + e].J; }
        wgs.push_back(w);
        ++team[i];
      }
      k += nt;
    }
  }

  void carve(psh::Arena& ar, bool assign) {
    PiBlock* blk = ar.take<PiBlock>(batch);
    PiTile* tl = ar.take<PiTile>(std::max<size_t>(tiles.size(), 1));
    float* v0 = ar.take<float>(std::max(max_n, 1));
    int* asym = ar.take<int>(std::max(batch, 1));
    // launch list of the resident passes: the team workgroups dealt to the 8 XCDs and padded
    // (layout_for_xcds): at most 8 * (ceil(total / 8) + largest team) entries per pass
    int big_team = 1;
    for (int i = 0; i < batch; ++i) big_team = std::max(big_team, team.empty() ? 1 : team[i]);
    PiTeamWG* wg = ar.take<PiTeamWG>(2 * wgs.size() + 16 * (size_t)big_team + 64);
    if (assign) {
      d_blocks = blk; d_tiles = tl; d_v0 = v0; d_asym = asym; d_wgs = wg;
      d_vn.clear(); d_P.clear();
    }
    char* region0 = ar.base ? ar.base + psh::align_up(ar.off, 256) : nullptr;
    for (int i = 0; i < batch; ++i) {
      const int t = (n_eff[i] + PT - 1) / PT;
      float* vn = ar.take<float>(std::max(t * PT, 1));
      // resident execution: 8-byte granules, two step parities of the partial slab + of y
      // + the abort granule + one placement granule per team member
      float* P = ar.take<float>(std::max(4 * (t * t + t) * PT + 2 + 2 * (team.empty() ? 1 : team[i]), 1));
      if (assign) { d_vn.push_back(vn); d_P.push_back(P); }
    }
    if (assign) { d_region = region0; region_bytes = ar.base ? (size_t)(ar.base + ar.off - region0) : 0; }
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
      pb.team = team[i];
    }
    std::vector<float> v0(std::max(max_n, 1));
    ps_power_iteration_v0(max_n, v0.data());
    PS_RC(psh::upload_async(st, d_blocks, h.data(), sizeof(PiBlock) * batch));
    PS_RC(psh::upload_async(st, d_v0, v0.data(), sizeof(float) * v0.size()));
    if (!tiles.empty())
      PS_RC(psh::upload_async(st, d_tiles, tiles.data(), sizeof(PiTile) * tiles.size()));
    // (the team workgroups are uploaded by enqueue(), in launch order)
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

  // Co-resident capacity of the resident kernel on this device (workgroups), 0 = unusable.
  static int resident_capacity(size_t dyn_lds) {
    static int cus = -1;
    if (cus < 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        cus = 0;
      else
        cus = prop.multiProcessorCount;
      (void)hipFuncSetAttribute((const void*)pi_resident_kernel<RNT>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    }
    int per_cu = 0;
    if (cus <= 0 || hipOccupancyMaxActiveBlocksPerMultiprocessor(
                        &per_cu, pi_resident_kernel<RNT>, 256, dyn_lds) != hipSuccess)
      return 0;
    // the occupancy API can answer one block per CU too high for kernels with ~100 SGPRs
    // (MI355X_MICROARCH.md, correctness boundaries): never exceed the launch bound
    const int bound = RNT >= 3 ? 2 : (RNT == 2 ? 3 : 4);
    return std::min(per_cu, bound) * cus;
  }

  // Process-wide health of the resident execution.  `expired` is pinned host memory that the
  // kernel counts into when a bounded wait runs out (a team mate that never became resident:
  // the launch size assumes an otherwise idle chip); after the first such event the process
  // stays on the streaming execution.  `collectives` is raised by the host around asynchronous
  // RCCL gathers (ps_collective_in_flight): their kernels hold CUs on another stream, so the
  // resident execution is not used while one is in flight.
  struct PiHealth {
    unsigned* expired = nullptr;
    std::atomic<int> collectives{0};
  };
  static PiHealth& health() {
    static PiHealth h;
    static std::once_flag once;
    std::call_once(once, [] {
      void* p = nullptr;
      if (hipHostMalloc(&p, 64, hipHostMallocMapped) == hipSuccess) {
        h.expired = static_cast<unsigned*>(p);
        *h.expired = 0;
      }
    });
    return h;
  }
  static unsigned expired_total() {
    PiHealth& h = health();
    return h.expired ? *reinterpret_cast<volatile unsigned*>(h.expired) : 0u;
  }
  // Per-call choices (ps_options.power_iteration / pi_timeout_ms, set by the drivers from
  // psh::Options): whether this call may use the resident execution at all, and the deadline of
  // a resident launch's waits.
  bool allow_resident = true;
  double timeout_ms = 5000.0;
  void set_options(const psh::Options& o) {
    allow_resident = o.power_iteration != PS_PI_STREAMING;
    timeout_ms = o.pi_timeout_ms;
  }
  unsigned long long timeout_ticks() const {   // of the 100 MHz constant clock
    return (unsigned long long)(std::max(timeout_ms, 0.0) * 1.0e5);   // 0: every wait counts as expired (tests)
  }

  // Process health only (expired waits so far, collectives in flight): what is shared between
  // callers is the CHIP, not a mode.
  static bool resident_enabled() {
    PiHealth& h = health();
    return h.expired != nullptr && *h.expired == 0 && h.collectives.load() == 0;
  }

  // Two resident launches on DIFFERENT streams could each hold part of the chip while their
  // teams wait for slots: the resident execution is used only when the previous resident
  // launch of this process went to the same stream or has completed.  acquire() claims the
  // chip for `st` (or refuses), release() records the completion event behind the launches.
  struct ResidentGate {
    std::mutex mu;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    bool any = false, pending = false;
  };
  static ResidentGate& gate() { static ResidentGate g; return g; }
  static bool resident_acquire(hipStream_t st) {
    ResidentGate& g = gate();
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.ev && hipEventCreateWithFlags(&g.ev, hipEventDisableTiming) != hipSuccess) return false;
    if (g.any && g.stream != st && (g.pending || hipEventQuery(g.ev) != hipSuccess)) return false;
    if (g.any && g.stream == st && g.pending) return false;  // same stream from two threads
    g.stream = st; g.any = true; g.pending = true;
    return true;
  }
  static void resident_release(hipStream_t st) {
    ResidentGate& g = gate();
    std::lock_guard<std::mutex> lk(g.mu);
    (void)hipEventRecord(g.ev, st);
    g.pending = false;
  }

  // Enqueues the whole iteration.  Resident execution (default): one launch per co-resident
  // pass, matrices in registers.  Streaming execution (PS_PI_STREAMING, teams larger than the
  // chip, or a resident launch in flight on another stream): 2 launches per step, fixed
  // count (data-dependent stops are taken on the device; stopped blocks' workgroups exit at
  // once).  Same arithmetic, bit-identical results.
  int enqueue(hipStream_t st, int num_iters, float tol) {
    const size_t shm = 0;
    const size_t red_shm = (size_t)((max_n + PT - 1) / PT) * PT * sizeof(float);  // <= 64 KB
    if (batch == 0) return 0;
    const size_t res_lds = 2 * red_shm;
    const int cap = (allow_resident && resident_enabled() && num_iters > 0 && res_lds <= 48 * 1024)
                        ? resident_capacity(res_lds) : 0;
    int biggest = 0;
    for (int i = 0; i < batch; ++i) biggest = std::max(biggest, team[i]);
    const bool resident = cap > 0 && biggest <= cap && resident_acquire(st);
    // granule tags must not match leftovers of an earlier call in the same workspace
    if (resident && region_bytes) {
      const hipError_t e = hipMemsetAsync(d_region, 0, region_bytes, st);
      if (e != hipSuccess) { resident_release(st); return (int)e; }
    }
    hipLaunchKernelGGL(pi_init_kernel, dim3(batch), dim3(256), 0, st, d_blocks, d_v0);
    if (resident) {
      // Passes of whole teams, in block order, at most `cap` workgroups each.  Inside a pass the
      // teams are dealt to the 8 XCDs (least loaded first, at most cap / 8 workgroups each) and the
      // per-XCD lists are interleaved -- launch index j * 8 + x runs on XCD x -- with no-op
      // entries where a list is shorter: a team's workgroups then share one L2 and hand off
      // through it (pi_publish).
      std::vector<PiTeamWG> launch;
      std::vector<std::pair<size_t, size_t>> passes;   // (first, count) into `launch`
      const PiTeamWG noop{-1, 0, 0, {0, 0, 0, 0}, {0, 0, 0, 0}, 0, 1};
      const int per_xcd = cap / psh::NXCD;
      size_t next = 0;
      while (next < wgs.size()) {
        if (team[wgs[next].block] > per_xcd) {
          // a team larger than one XCD's share of the resident slots: a pass of its own in plain
          // launch order (spread over the XCDs; it hands off through memory as before)
          const int tm = team[wgs[next].block];
          passes.emplace_back(launch.size(), (size_t)tm);
          launch.insert(launch.end(), wgs.begin() + next, wgs.begin() + next + tm);
          next += tm;
          continue;
        }
        std::vector<PiTeamWG> bins[psh::NXCD];
        int used = 0;
        while (next < wgs.size()) {
          const int tm = team[wgs[next].block];
          if (tm > per_xcd || used + tm > cap) break;
          int best = 0;
          for (int x = 1; x < psh::NXCD; ++x)
            if (bins[x].size() < bins[best].size()) best = x;
          if ((int)bins[best].size() + tm > per_xcd) break;
          for (int m = 0; m < tm; ++m) bins[best].push_back(wgs[next + m]);
          used += tm;
          next += tm;
        }
        std::vector<PiTeamWG> pass;
        psh::interleave_xcd_lists(bins, noop, pass);
        passes.emplace_back(launch.size(), pass.size());
        launch.insert(launch.end(), pass.begin(), pass.end());
      }
      if (launch.size() > 2 * wgs.size() + 16 * (size_t)biggest + 64) { resident_release(st); return PS_EINTERNAL; }
      {
        const int rc = psh::upload_async(st, d_wgs, launch.data(), sizeof(PiTeamWG) * launch.size());
        if (rc) { resident_release(st); return rc; }
      }
      for (const auto& ps : passes)
        hipLaunchKernelGGL(pi_resident_kernel<RNT>, dim3((unsigned)ps.second), dim3(256), res_lds, st,
                           d_blocks, d_wgs + ps.first, num_iters, tol, timeout_ticks(),
                           health().expired);
      resident_release(st);
      PS_LAUNCH_CHECK();
      return 0;
    }
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
