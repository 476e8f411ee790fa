// power_iter.hip.h — batched power iteration (reference: power_iteration,
// DS:595-652) as one launch per step over every (block, 32-row chunk).
//
// HBM/L2-bound: one step reads each matrix once (n*n*4 bytes per block).  The
// loop of DS:649 is data dependent (stop when |s_new - s| <= tol), so each
// workgroup re-derives the block's stop decision from the previous step's
// partial sums in a fixed order: every workgroup of a block takes the same
// decision without any inter-workgroup communication inside a launch, and the
// result is bit-reproducible (no float atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "gemm_core.hip.h"

namespace psk {

constexpr int PI_ROWS = 32;  // rows per workgroup

struct PiBlock {
  const float* a;
  int lda;
  int n;            // effective size (rows/cols >= n are padding and ignored)
  int vec_ok;       // float4 row loads are legal
  float* v[2];      // ping-pong iterate, >= n floats each; v[0] starts as v0[:n]
  float* partial[2];  // per-chunk partial sums of v.(A v), by step parity
  int nchunk;
  float s_val[2];   // s of steps (i-1), (i-2) by parity
  int stop_iter;    // -1 while running, else the step after which the loop ended
  float lambda;     // s_out
  int iters;        // steps executed
};

struct PiChunk {
  int block;
  int chunk;  // index inside the block; rows [chunk*PI_ROWS, ...)
};

// sum of x[0..cnt) in a fixed order, computed by wavefront 0; result broadcast
// through LDS slot `bcast`.
__device__ inline float fixed_order_sum_wave0(const float* x, int cnt, int tid,
                                              float* bcast) {
  if (tid < 64) {
    float s = 0.f;
    for (int j = tid; j < cnt; j += 64) s += x[j];
    s = wave_sum_f32(s);
    if (tid == 0) *bcast = s;
  }
  __syncthreads();
  return *bcast;
}

static __global__ __launch_bounds__(256) void pi_step_kernel(PiBlock* blocks,
                                                      const PiChunk* chunks,
                                                      int iter, float tol) {
  extern __shared__ __align__(16) float pi_smem[];  // [n] normalised v, then scratch
  __shared__ float red[8];
  __shared__ int s_stop;
  const PiChunk ch = chunks[blockIdx.x];
  PiBlock* pb = &blocks[ch.block];
  const int tid = threadIdx.x;
  const int n = pb->n;
  // Another workgroup of this block may be recording the stop right now: take
  // the decision once per workgroup so that it is uniform.
  if (tid == 0) s_stop = pb->stop_iter >= 0 ? 1 : 0;
  __syncthreads();
  if (s_stop) return;

  if (iter >= 1) {
    // s of the previous step, and the run_step predicate of DS:639.
    const float s_cur =
        fixed_order_sum_wave0(pb->partial[(iter - 1) & 1], pb->nchunk, tid, &red[7]);
    if (tid == 0) {
      const float s_prev = iter >= 2 ? pb->s_val[iter & 1] : 0.f;
      const bool run = fabsf(s_cur - s_prev) > tol;
      s_stop = run ? 0 : 1;
      if (ch.chunk == 0) {
        pb->s_val[(iter - 1) & 1] = s_cur;
        if (!run) {
          pb->lambda = s_cur;
          pb->iters = iter;
          pb->stop_iter = iter - 1;
        }
      }
    }
    __syncthreads();
    if (s_stop) return;
  }

  const float* v = pb->v[iter & 1];
  float* vnext = pb->v[(iter + 1) & 1];

  // ||v||, same order in every workgroup of the block (DS:634).
  float ss = 0.f;
  for (int j = tid; j < n; j += 256) ss += v[j] * v[j];
  ss = wave_sum_f32(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float nrm = sqrtf(((red[0] + red[1]) + red[2]) + red[3]);
  for (int j = tid; j < n; j += 256) pi_smem[j] = v[j] / nrm;
  __syncthreads();

  // rows of this chunk: A v (DS:636) and the partial of v.(A v) (DS:637).
  const int wave = tid >> 6, lane = tid & 63;
  float spart = 0.f;
  const int row_end = min(n, (ch.chunk + 1) * PI_ROWS);
  for (int r = ch.chunk * PI_ROWS + wave; r < row_end; r += 4) {
    const float* arow = pb->a + (int64_t)r * pb->lda;
    float acc = 0.f;
    if (pb->vec_ok) {
      const int n4 = n & ~3;
      for (int j = lane * 4; j < n4; j += 256) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(arow + j);
        const f32x4 y = *reinterpret_cast<const f32x4*>(pi_smem + j);
        acc += x[0] * y[0];
        acc += x[1] * y[1];
        acc += x[2] * y[2];
        acc += x[3] * y[3];
      }
      for (int j = n4 + lane; j < n; j += 64) acc += arow[j] * pi_smem[j];
    } else {
      for (int j = lane; j < n; j += 64) acc += arow[j] * pi_smem[j];
    }
    acc = wave_sum_f32(acc);
    if (lane == 0) {
      vnext[r] = acc;
      spart += pi_smem[r] * acc;
    }
  }
  __syncthreads();
  if (lane == 0) red[wave] = spart;
  __syncthreads();
  if (tid == 0)
    pb->partial[iter & 1][ch.chunk] = ((red[0] + red[1]) + red[2]) + red[3];
}

// One workgroup per block: closes the loop (DS:649-652) and optionally writes
// the normalised vector.
static __global__ __launch_bounds__(256) void pi_final_kernel(PiBlock* blocks, int num_iters,
                                                       float* out_lambda,
                                                       int* out_iters, float* out_v,
                                                       int ldv) {
  __shared__ float red[8];
  PiBlock* pb = &blocks[blockIdx.x];
  const int tid = threadIdx.x;
  const int n = pb->n;
  if (pb->stop_iter < 0) {
    float s;
    if (n == 0) {
      s = __uint_as_float(0x7fc00000u);  // v0 masked to zero: 0/0 (DS:634)
      if (tid == 0) { pb->lambda = s; pb->iters = 1; pb->stop_iter = 0; }
    } else {
      s = fixed_order_sum_wave0(pb->partial[(num_iters - 1) & 1], pb->nchunk, tid,
                                &red[7]);
      if (tid == 0) { pb->lambda = s; pb->iters = num_iters; pb->stop_iter = num_iters - 1; }
    }
  }
  __syncthreads();
  if (tid == 0) {
    if (out_lambda) out_lambda[blockIdx.x] = pb->lambda;
    if (out_iters) out_iters[blockIdx.x] = pb->iters;
  }
  if (out_v != nullptr && n > 0) {
    const float* v = pb->v[pb->iters & 1];
    float ss = 0.f;
    for (int j = tid; j < n; j += 256) ss += v[j] * v[j];
    ss = wave_sum_f32(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf(((red[0] + red[1]) + red[2]) + red[3]);
    for (int j = tid; j < n; j += 256)
      out_v[(int64_t)blockIdx.x * ldv + j] = v[j] / nrm;  // DS:651
  }
}

}  // namespace psk
