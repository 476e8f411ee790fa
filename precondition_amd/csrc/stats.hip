// stats.hip — Kronecker-factor statistics accumulation
//   S <- w1*S + w2*tensordot(g, g, axes != axis)      (gram_weighted_update, DS:1440-1470)
// for every (block, axis) pair of a parameter tree in one launch per operand
// layout (the reference unrolls this loop in Python at trace time, DS:1582-1590).
//
// The Gram matrix is X X^T with X read straight out of the gradient block (no
// transposed copy): "k-contiguous" rows go through the KC LDS image for both
// operands, "d-contiguous" columns through the MC image for both (gemm_core).
// MFMA-bound: arithmetic intensity ~ d/4 flop per byte of g for one axis.
// Epilogue: out = fl(fl(w1*old) + fl(w2*gram)), the rounding sequence of DS:1470.
// X X^T is symmetric and tile (j,i) of it is the bitwise transpose of tile (i,j) (same
// products, same k order), so only the tiles with tm <= tn are computed; the strictly
// upper ones are mirrored through an LDS transpose, with `old` read at the mirrored
// position (a caller-supplied asymmetric `old` therefore stays asymmetric exactly as in
// the reference).  This removes (T-1)/(2T) of the MFMA work, T = tiles per side.
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "gemm_core.hip.h"

namespace psk {

constexpr int SBK = 32;  // K-tile; two register sets of prefetch (gemm_core DEEP), 2 WG/CU
constexpr int STATS_LDS_BYTES = SmemCfg<SBK>::TOTAL * (int)sizeof(float);

struct StatsTask {
  const float* g;
  const float* sin;
  float* sout;
  int64_t seg_stride;
  int d, k, ld, lds, nseg, vec;
  int fast;    // interior tiles may take the unguarded K loop (aligned g, k % SBK == 0)
  int svec;    // the statistic allows 16-byte accesses (aligned pointers, lds % 4 == 0)
  int layout;  // KC / MC
};

struct StatsTile {
  int task;
  short tm, tn;
};

__device__ __forceinline__ float stats_blend(float w1, float old, float w2, float g) {
  return __fadd_rn(__fmul_rn(w1, old), __fmul_rn(w2, g));  // DS:1470's rounding sequence
}

// out[tn-tile rows, tm-tile cols] = w1*old + w2*acc^T, in two 64-row halves through LDS
// (stride 129: conflict-free both ways) so that the global accesses are 256-byte runs.
// Bounds-checked: edge tiles and unaligned statistics.
__device__ inline void stats_store_mirror(const f32x16 (&acc)[2][2], float* smem,
                                          const StatsTask& tk, int tm, int tn, float w1,
                                          float w2) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int TLD = 129;
  const int row0 = tn * TILE, col0 = tm * TILE;
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
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
      const int row = row0 + c, col = col0 + h * 64 + lr;
      if (row < tk.d && col < tk.d) {
        const int64_t o = (int64_t)row * tk.lds + col;
        gstore1(tk.sout + o, stats_blend(w1, gload1(tk.sin + o), w2, smem[lr * TLD + c]));
      }
    }
    __syncthreads();
  }
}

// The same for an interior tile of a 16-byte aligned statistic: the transposed image T[c][r]
// (gemm_core store_tile_transposed_v4's layout: stride 132, ds_write/read_b128) and 16-byte
// global loads of `old` / stores of the result, 512-byte runs per 32 lanes.
__device__ inline void stats_store_mirror_v4(const f32x16 (&acc)[2][2], float* smem,
                                             const StatsTask& tk, int tm_, int tn_, float w1,
                                             float w2) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int TLD = 132;
  const int row0 = tn_ * TILE, col0 = tm_ * TILE;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float* trow = smem + (wn * 32 + (lane & 31)) * TLD + wm * 64 + 4 * (lane >> 5);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[tm][h][4 * g + 0], acc[tm][h][4 * g + 1], acc[tm][h][4 * g + 2],
                   acc[tm][h][4 * g + 3]};
        *reinterpret_cast<f32x4*>(trow + tm * 32 + 8 * g) = v;
      }
    __syncthreads();
    const int r4 = (tid & 31) * 4;
    f32x4 old[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ci = (tid >> 5) + 8 * k;
      const int c = (ci >> 5) * 64 + h * 32 + (ci & 31);
      old[k] = gload4(tk.sin + (int64_t)(row0 + c) * tk.lds + col0 + r4);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ci = (tid >> 5) + 8 * k;
      const int c = (ci >> 5) * 64 + h * 32 + (ci & 31);
      const f32x4 v = *reinterpret_cast<const f32x4*>(smem + ci * TLD + r4);
      f32x4 o = {stats_blend(w1, old[k][0], w2, v[0]), stats_blend(w1, old[k][1], w2, v[1]),
                 stats_blend(w1, old[k][2], w2, v[2]), stats_blend(w1, old[k][3], w2, v[3])};
      *(f32x4 PS_GLOBAL*)(tk.sout + (int64_t)(row0 + c) * tk.lds + col0 + r4) = o;
    }
    __syncthreads();
  }
}

template <int LAYOUT, bool FAST>
__device__ __forceinline__ void stats_tile(const StatsTask& tk, int tm, int tn, float w1,
                                           float w2, float* smem) {
  f32x16 acc[2][2];
  zero_acc(acc);
  // nseg == 1 (every matrix-shaped block) outside the segment loop: the loop around the
  // two-register-set pipeline costs ~100 spilled VGPRs, which only rank>2 blocks then pay
  if (tk.nseg == 1) {
    Operand A{tk.g, tk.ld, tm * TILE, tk.d, tk.k, tk.vec != 0};
    Operand B{tk.g, tk.ld, tn * TILE, tk.d, tk.k, tk.vec != 0};
    gemm_tile_accum<LAYOUT, LAYOUT, SBK, !FAST, FAST>(A, B, tk.k, smem, acc);
  } else {
#pragma unroll 1
    for (int s = 0; s < tk.nseg; ++s) {
      const float* base = tk.g + (int64_t)s * tk.seg_stride;
      Operand A{base, tk.ld, tm * TILE, tk.d, tk.k, tk.vec != 0};
      Operand B{base, tk.ld, tn * TILE, tk.d, tk.k, tk.vec != 0};
      gemm_tile_accum<LAYOUT, LAYOUT, SBK, !FAST, false>(A, B, tk.k, smem, acc);
    }
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  if (FAST) {
    // wave-uniform block bases + one per-lane offset (newton.hip's epilogue addressing)
    const int lane_off = 4 * (lane >> 5) * tk.lds + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t b = (int64_t)(tm * TILE + wm * 64 + i * 32) * tk.lds + tn * TILE +
                          wn * 64 + j * 32;
        const float* oblk = tk.sin + b;
        float* nblk = tk.sout + b;
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          old[r] = gload1(oblk + (int64_t)((r & 3) + 8 * (r >> 2)) * tk.lds + lane_off);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          gstore1(nblk + (int64_t)((r & 3) + 8 * (r >> 2)) * tk.lds + lane_off,
                  stats_blend(w1, old[r], w2, acc[i][j][r]));
      }
    if (tm != tn) stats_store_mirror_v4(acc, smem, tk, tm, tn, w1, w2);
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tm * TILE + acc_row(wm, i, r, lane);
        const int col = tn * TILE + acc_col(wn, j, lane);
        if (row < tk.d && col < tk.d) {
          const int64_t o = (int64_t)row * tk.lds + col;
          gstore1(tk.sout + o, stats_blend(w1, gload1(tk.sin + o), w2, acc[i][j][r]));
        }
      }
  if (tm != tn) stats_store_mirror(acc, smem, tk, tm, tn, w1, w2);
}

// A tile takes the fast path when it is interior, the contraction length is a whole number
// of K-tiles and the host found every pointer / leading dimension 16-byte aligned.
__device__ __forceinline__ bool stats_tile_is_fast(const StatsTask& tk, int tm, int tn) {
  return tk.fast != 0 && tk.svec != 0 && (tn + 1) * TILE <= tk.d && (tm + 1) * TILE <= tk.d;
}

template <int LAYOUT>
__device__ __forceinline__ void stats_tile_any(const StatsTask& tk, int tm, int tn, float w1,
                                               float w2, float* smem) {
  if (stats_tile_is_fast(tk, tm, tn))
    stats_tile<LAYOUT, true>(tk, tm, tn, w1, w2, smem);
  else
    stats_tile<LAYOUT, false>(tk, tm, tn, w1, w2, smem);
}

// Statistic of a vector block (k = 1: biases, scales): S = w1*S + w2 * g g^T is one product per
// element, fl(g_i * g_j) -- exactly what the MFMA path returns for a single k -- so the tile is
// streamed by the VALU (16-byte accesses where the statistic allows them) without the LDS
// pipeline.  Writes the 128x128 tile at (rt, ct); the caller passes both orders of an
// off-diagonal pair.  A ViT-B tree holds 149 such statistics, 0.85 GB of read + write.
__device__ inline void stats_vector_tile(const StatsTask& tk, int rt, int ct, float w1,
                                         float w2) {
  const int tid = threadIdx.x;
  const int c = ct * TILE + (tid & 31) * 4;
  const int estride = tk.layout == KC ? tk.ld : 1;  // element i of the vector
  float gc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    gc[e] = c + e < tk.d ? gload1(tk.g + (int64_t)(c + e) * estride) : 0.f;
  const bool v4 = tk.svec != 0 && c + 3 < tk.d;
  if (v4 && (rt + 1) * TILE <= tk.d) {
    // whole rows in range: 8 loads in flight per lane (the workgroup count per CU is set by
    // the LDS of the matrix path, so the memory parallelism has to come from the lane)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 old[8];
      float gr[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int r = rt * TILE + (half * 8 + p) * 8 + (tid >> 5);
        gr[p] = gload1(tk.g + (int64_t)r * estride);
        old[p] = gload4(tk.sin + (int64_t)r * tk.lds + c);
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int r = rt * TILE + (half * 8 + p) * 8 + (tid >> 5);
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          out[e] = stats_blend(w1, old[p][e], w2, __fmul_rn(gr[p], gc[e]));
        *(f32x4 PS_GLOBAL*)(tk.sout + (int64_t)r * tk.lds + c) = out;
      }
    }
    return;
  }
#pragma unroll 1
  for (int p = 0; p < 16; ++p) {
    const int r = rt * TILE + p * 8 + (tid >> 5);
    if (r >= tk.d) break;
    const float gr = gload1(tk.g + (int64_t)r * estride);
    const int64_t o = (int64_t)r * tk.lds + c;
    if (v4) {
      const f32x4 old = gload4(tk.sin + o);
      f32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = stats_blend(w1, old[e], w2, __fmul_rn(gr, gc[e]));
      *(f32x4 PS_GLOBAL*)(tk.sout + o) = out;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c + e < tk.d)
          gstore1(tk.sout + o + e,
                  stats_blend(w1, gload1(tk.sin + o + e), w2, __fmul_rn(gr, gc[e])));
    }
  }
}

// One launch per parameter tree: every tile of every statistic, both operand layouts and the
// vector statistics (the branch is uniform per workgroup).  The tile list arrives in hardware
// dispatch order: workgroup b runs on XCD b % 8, and the host has dealt the statistics over
// the 8 XCDs by cost (make_tile_list), padding the shorter lists with task = -1.
__global__ __launch_bounds__(256, 2) void stats_grouped_kernel(const StatsTask* tasks,
                                                               const StatsTile* tiles,
                                                               int ntiles, float w1,
                                                               float w2) {
  extern __shared__ __align__(16) float smem[];
  const StatsTile te = tiles[blockIdx.x];
  if (te.task < 0) return;
  const StatsTask tk = tasks[te.task];
  if (tk.k == 1 && tk.nseg == 1) {
    stats_vector_tile(tk, te.tm, te.tn, w1, w2);
    if (te.tm != te.tn) stats_vector_tile(tk, te.tn, te.tm, w1, w2);
  } else if (tk.layout == KC) {
    stats_tile_any<KC>(tk, te.tm, te.tn, w1, w2, smem);
  } else {
    stats_tile_any<MC>(tk, te.tm, te.tn, w1, w2, smem);
  }
}

template <int LAYOUT>
__global__ __launch_bounds__(256, 2) void stats_single_kernel(StatsTask tk, int tiles_n,
                                                              float w1, float w2) {
  extern __shared__ __align__(16) float smem[];
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  if (t / tiles_n > t % tiles_n) return;  // lower tiles are written by their mirrors
  stats_tile_any<LAYOUT>(tk, t / tiles_n, t % tiles_n, w1, w2, smem);
}

}  // namespace psk

using namespace psk;

namespace {

using psh::NXCD;

bool to_task(const ps_stats_desc& d, StatsTask& t) {
  if (!d.g || !d.stat_in || !d.stat_out || d.d < 1 || d.k < 1 || d.nseg < 1 ||
      (d.layout != 0 && d.layout != 1) || d.lds < d.d || d.ld < 1 ||
      d.ld > 0x7fffffff || d.lds > 0x7fffffff)
    return false;
  if (d.layout == 0 && d.ld < d.k) return false;
  if (d.layout == 1 && d.ld < d.d) return false;
  t.g = d.g; t.sin = d.stat_in; t.sout = d.stat_out;
  t.seg_stride = d.seg_stride;
  t.d = d.d; t.k = d.k; t.ld = (int)d.ld; t.lds = (int)d.lds; t.nseg = d.nseg;
  t.vec = ((uintptr_t)d.g % 16 == 0) && (d.ld % 4 == 0) && (d.seg_stride % 4 == 0);
  t.fast = t.vec && d.k % SBK == 0;
  t.svec = (uintptr_t)d.stat_in % 16 == 0 && (uintptr_t)d.stat_out % 16 == 0 && d.lds % 4 == 0;
  t.layout = d.layout;
  return true;
}

size_t grouped_bytes(const ps_stats_desc* desc, int count) {
  size_t tiles = 0;
  for (int i = 0; i < count; ++i) {
    const size_t t = (desc[i].d + TILE - 1) / TILE;
    tiles += t * t;
  }
  // the dealt list is padded to 8 x the longest per-XCD list (<= 8 x tiles)
  return psh::align_up(sizeof(StatsTask) * count, 256) +
         psh::align_up(sizeof(StatsTile) * tiles * NXCD, 256) + 1024;
}

// Tile list in dispatch order (common.h deal_to_xcds): per-tile cost = K-tiles of the
// contraction + an epilogue term; a vector statistic's tile is a short stream.
void make_tile_list(const std::vector<StatsTask>& tasks, std::vector<StatsTile>& out) {
  std::vector<int> ntile(tasks.size());
  std::vector<int64_t> cost(tasks.size());
  for (size_t i = 0; i < tasks.size(); ++i) {
    const StatsTask& t = tasks[i];
    const int nt = (t.d + TILE - 1) / TILE;
    ntile[i] = nt * (nt + 1) / 2;
    const bool vec = t.k == 1 && t.nseg == 1;
    cost[i] = vec ? 3 : ((int64_t)t.k + SBK - 1) / SBK * t.nseg + 6;
  }
  std::vector<psh::DealUnit> units[psh::NXCD];
  psh::deal_to_xcds(ntile, cost, units);
  std::vector<StatsTile> lists[psh::NXCD];
  for (int x = 0; x < psh::NXCD; ++x)
    for (const psh::DealUnit& un : units[x]) {
      const int nt = (tasks[un.task].d + TILE - 1) / TILE;
      int idx = 0;
      for (int tm = 0; tm < nt; ++tm)
        for (int tn = tm; tn < nt; ++tn, ++idx)
          if (idx >= un.first && idx < un.first + un.count)
            lists[x].push_back({un.task, (short)tm, (short)tn});
    }
  psh::interleave_xcd_lists(lists, StatsTile{-1, 0, 0}, out);
}

template <typename K>
void allow_lds(K kernel) {
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            STATS_LDS_BYTES);
}

void stats_lds_once() {
  static const bool done = [] {
    allow_lds(stats_grouped_kernel);
    allow_lds(stats_single_kernel<KC>);
    allow_lds(stats_single_kernel<MC>);
    return true;
  }();
  (void)done;
}

}  // namespace

extern "C" size_t ps_stats_update_grouped_workspace_bytes(const ps_stats_desc* desc,
                                                          int count) {
  if (!desc || count <= 0) return 0;
  return grouped_bytes(desc, count);
}

extern "C" int ps_stats_update_grouped_f32(void* stream, const ps_stats_desc* desc,
                                           int count, float w1, float w2,
                                           void* workspace, size_t workspace_bytes) {
  PS_DEVICE_CHECK();
  if (!desc || count <= 0 || !workspace) return PS_EINVAL;
  if (workspace_bytes < grouped_bytes(desc, count)) return PS_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  stats_lds_once();
  std::vector<StatsTask> tasks(count);
  for (int i = 0; i < count; ++i)
    if (!to_task(desc[i], tasks[i])) return PS_EINVAL;
  std::vector<StatsTile> tiles;
  make_tile_list(tasks, tiles);
  // one staging buffer, one H2D copy: [tasks][tiles]
  const size_t task_bytes = psh::align_up(sizeof(StatsTask) * tasks.size(), 256);
  const size_t tile_bytes = sizeof(StatsTile) * tiles.size();
  psh::Arena ar(workspace, workspace_bytes);
  char* d_plan = ar.take<char>(task_bytes + tile_bytes);
  if (ar.overflow) return PS_EWORKSPACE;
  std::vector<char> plan(task_bytes + tile_bytes);
  memcpy(plan.data(), tasks.data(), sizeof(StatsTask) * tasks.size());
  memcpy(plan.data() + task_bytes, tiles.data(), tile_bytes);
  PS_RC(psh::upload_async(st, d_plan, plan.data(), plan.size()));
  const int nt = (int)tiles.size();
  hipLaunchKernelGGL(stats_grouped_kernel, dim3(nt), dim3(256), STATS_LDS_BYTES, st,
                     (const StatsTask*)d_plan, (const StatsTile*)(d_plan + task_bytes), nt, w1,
                     w2);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_stats_update_f32(void* stream, const float* g, int64_t rows,
                                   int64_t cols, int64_t ldg, int axis,
                                   const float* stat_in, float* stat_out, int64_t lds,
                                   float w1, float w2) {
  PS_DEVICE_CHECK();
  if (rows < 1 || cols < 1 || (axis != 0 && axis != 1) || rows > 0x7fffffff ||
      cols > 0x7fffffff || ldg < cols)
    return PS_EINVAL;
  ps_stats_desc d;
  memset(&d, 0, sizeof(d));
  d.g = g; d.layout = axis; d.d = (int32_t)(axis == 0 ? rows : cols);
  d.k = (int32_t)(axis == 0 ? cols : rows); d.nseg = 1; d.ld = ldg; d.seg_stride = 0;
  d.stat_in = stat_in; d.stat_out = stat_out; d.lds = lds;
  StatsTask t;
  if (!to_task(d, t)) return PS_EINVAL;
  stats_lds_once();
  const int nt = (t.d + TILE - 1) / TILE;
  if (axis == 0)
    hipLaunchKernelGGL(stats_single_kernel<KC>, dim3(nt * nt), dim3(256), STATS_LDS_BYTES,
                       (hipStream_t)stream, t, nt, w1, w2);
  else
    hipLaunchKernelGGL(stats_single_kernel<MC>, dim3(nt * nt), dim3(256), STATS_LDS_BYTES,
                       (hipStream_t)stream, t, nt, w1, w2);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
