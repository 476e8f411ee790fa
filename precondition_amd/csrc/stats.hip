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

#include <vector>

#include "common.h"
#include "gemm_core.hip.h"

namespace psk {

constexpr int SBK = 16;

struct StatsTask {
  const float* g;
  const float* sin;
  float* sout;
  int64_t seg_stride;
  int d, k, ld, lds, nseg, vec;
};

struct StatsTile {
  int task;
  short tm, tn;
};

// out[tn-tile rows, tm-tile cols] = w1*old + w2*acc^T, in two 64-row halves through LDS
// (stride 129: conflict-free both ways) so that the global accesses are 256-byte runs.
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
        gstore1(tk.sout + o, __fadd_rn(__fmul_rn(w1, gload1(tk.sin + o)),
                                       __fmul_rn(w2, smem[lr * TLD + c])));
      }
    }
    __syncthreads();
  }
}

template <int LAYOUT>
__device__ inline void stats_tile(const StatsTask& tk, int tm, int tn, float w1, float w2,
                                  float* smem) {
  f32x16 acc[2][2];
  zero_acc(acc);
  for (int s = 0; s < tk.nseg; ++s) {
    const float* base = tk.g + (int64_t)s * tk.seg_stride;
    Operand A{base, tk.ld, tm * TILE, tk.d, tk.k, tk.vec != 0};
    Operand B{base, tk.ld, tn * TILE, tk.d, tk.k, tk.vec != 0};
    gemm_tile_accum<LAYOUT, LAYOUT, SBK, true>(A, B, tk.k, smem, acc);
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
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
          gstore1(tk.sout + o, __fadd_rn(__fmul_rn(w1, gload1(tk.sin + o)),
                                         __fmul_rn(w2, acc[i][j][r])));
        }
      }
  if (tm != tn) stats_store_mirror(acc, smem, tk, tm, tn, w1, w2);
}

template <int LAYOUT>
__global__ __launch_bounds__(256, 2) void stats_grouped_kernel(const StatsTask* tasks,
                                                               const StatsTile* tiles,
                                                               int ntiles, float w1,
                                                               float w2) {
  __shared__ __align__(16) float smem[SmemCfg<SBK>::TOTAL];
  const StatsTile te = tiles[xcd_remap(blockIdx.x, ntiles)];
  const StatsTask tk = tasks[te.task];
  stats_tile<LAYOUT>(tk, te.tm, te.tn, w1, w2, smem);
}

template <int LAYOUT>
__global__ __launch_bounds__(256, 2) void stats_single_kernel(StatsTask tk, int tiles_n,
                                                              float w1, float w2) {
  __shared__ __align__(16) float smem[SmemCfg<SBK>::TOTAL];
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  if (t / tiles_n > t % tiles_n) return;  // lower tiles are written by their mirrors
  stats_tile<LAYOUT>(tk, t / tiles_n, t % tiles_n, w1, w2, smem);
}

}  // namespace psk

using namespace psk;

namespace {

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
  return true;
}

size_t grouped_bytes(const ps_stats_desc* desc, int count) {
  size_t tiles = 0;
  for (int i = 0; i < count; ++i) {
    const size_t t = (desc[i].d + TILE - 1) / TILE;
    tiles += t * t;
  }
  return psh::align_up(sizeof(StatsTask) * count, 256) +
         psh::align_up(sizeof(StatsTile) * tiles, 256) + 1024;
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
  std::vector<StatsTask> tasks[2];
  std::vector<StatsTile> tiles[2];
  for (int i = 0; i < count; ++i) {
    StatsTask t;
    if (!to_task(desc[i], t)) return PS_EINVAL;
    const int L = desc[i].layout;
    const int id = (int)tasks[L].size();
    tasks[L].push_back(t);
    const int nt = (t.d + TILE - 1) / TILE;
    for (int tm = 0; tm < nt; ++tm)
      for (int tn = tm; tn < nt; ++tn) tiles[L].push_back({id, (short)tm, (short)tn});
  }
  psh::Arena ar(workspace, workspace_bytes);
  StatsTask* d_tasks[2];
  StatsTile* d_tiles[2];
  for (int L = 0; L < 2; ++L) {
    d_tasks[L] = ar.take<StatsTask>(tasks[L].size());
    d_tiles[L] = ar.take<StatsTile>(tiles[L].size());
  }
  if (ar.overflow) return PS_EWORKSPACE;
  for (int L = 0; L < 2; ++L) {
    if (tasks[L].empty()) continue;
    PS_RC(psh::upload_async(st, d_tasks[L], tasks[L].data(), sizeof(StatsTask) * tasks[L].size()));
    PS_RC(psh::upload_async(st, d_tiles[L], tiles[L].data(), sizeof(StatsTile) * tiles[L].size()));
  }
  if (!tasks[0].empty()) {
    const int nt = (int)tiles[0].size();
    hipLaunchKernelGGL(stats_grouped_kernel<KC>, dim3(nt), dim3(256), 0, st, d_tasks[0],
                       d_tiles[0], nt, w1, w2);
  }
  if (!tasks[1].empty()) {
    const int nt = (int)tiles[1].size();
    hipLaunchKernelGGL(stats_grouped_kernel<MC>, dim3(nt), dim3(256), 0, st, d_tasks[1],
                       d_tiles[1], nt, w1, w2);
  }
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
  const int nt = (t.d + TILE - 1) / TILE;
  if (axis == 0)
    hipLaunchKernelGGL(stats_single_kernel<KC>, dim3(nt * nt), dim3(256), 0,
                       (hipStream_t)stream, t, nt, w1, w2);
  else
    hipLaunchKernelGGL(stats_single_kernel<MC>, dim3(nt * nt), dim3(256), 0,
                       (hipStream_t)stream, t, nt, w1, w2);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
