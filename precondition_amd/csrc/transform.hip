// transform.hip — fused, multi-tensor form of _transform_grad (reference DS:3496-3625):
// grafting (SGD / Adagrad / RMSProp (+normalised, +clip) / sqrt-n / none), norm
// matching of the preconditioned gradient, coupled or decoupled weight decay,
// momentum / Nesterov, for EVERY parameter of the tree in three launches.
//
// HBM-bound elementwise work with two to three per-parameter norms.  The norms
// are deterministic: each 4096-element chunk writes partial sums to a slab and
// every consumer workgroup re-reduces its parameter's partials in a fixed order
// (no float atomics).  Algorithmic bytes per element: pass A reads grad + pgrad
// (8 B), pass B reads grad + diag (8 B, only for Adagrad/RMSProp grafting), pass C
// reads up to 6 and writes up to 4 arrays (<= 40 B).
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "gemm_core.hip.h"

namespace psk {

constexpr int TCHUNK = 4096;
constexpr float T_EPS = 1e-25f;  // DS:41

enum Graft { G_NONE = 0, G_SGD = 1, G_ADAGRAD = 2, G_RMSPROP = 3, G_RMSPROP_N = 4,
             G_SQRT_N = 5, G_ADAGRAD_N = 6 };

struct TParam {
  const float* grad;
  const float* pgrad;   // preconditioned gradient; null => parameter skipped (DS:3557-3561)
  const float* param;   // may be null when weight_decay == 0
  const float* diag_in; // null unless Adagrad/RMSProp grafting
  float* diag_out;
  const float* mom_in;
  float* mom_out;
  const float* dmom_in;
  float* dmom_out;
  float* upd_out;
  int64_t numel;
  int chunk0, nchunks;
};

struct TChunk { int param; int idx; };

// chunk -> (parameter, chunk within it) from the per-parameter prefix chunk0: a binary
// search per workgroup instead of a host-built table of one entry per 4096 elements
// (21k entries for a ViT-B tree, rebuilt and uploaded every step).  Parameters without
// elements add no chunks; equal keys resolve to the last one, which owns the chunk.
__device__ inline TChunk find_chunk(const TParam* params, int count, int chunk) {
  int lo = 0, hi = count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (params[mid].chunk0 <= chunk) lo = mid; else hi = mid - 1;
  }
  return {lo, chunk - params[lo].chunk0};
}

__device__ inline bool has_diag(int g) {
  return g == G_ADAGRAD || g == G_RMSPROP || g == G_RMSPROP_N || g == G_ADAGRAD_N;
}
__device__ inline bool normalized(int g) { return g == G_RMSPROP_N || g == G_ADAGRAD_N; }

__device__ inline float block_sum(float v, float* red) {
  v = wave_sum_f32(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((red[0] + red[1]) + red[2]) + red[3];
}

// fixed-order sum of one slab column over a parameter's chunks
__device__ inline float param_sum(const float* slab, int col, const TParam& p, float* red) {
  float s = 0.f;
  for (int c = threadIdx.x; c < p.nchunks; c += 256) s += slab[(int64_t)(p.chunk0 + c) * 4 + col];
  return block_sum(s, red);
}

// pass A: slab[.][0] = sum (graft base)^2 ; slab[.][1] = sum pgrad^2
__global__ __launch_bounds__(256) void transform_pass_a(const TParam* params, int count,
                                                        float* slab, ps_transform_config cfg) {
  __shared__ float red[4];
  const TChunk ch = find_chunk(params, count, blockIdx.x);
  const TParam p = params[ch.param];
  const int64_t lo = (int64_t)ch.idx * TCHUNK;
  const int64_t hi = lo + TCHUNK < p.numel ? lo + TCHUNK : p.numel;
  float s0 = 0.f, s1 = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float g = gload1(p.grad + i);
    float base = g;
    if (cfg.graft_type == G_SQRT_N) base = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
    s0 += base * base;
    if (p.pgrad) { const float q = gload1(p.pgrad + i); s1 += q * q; }
  }
  s0 = block_sum(s0, red);
  s1 = block_sum(s1, red);
  if (threadIdx.x == 0) {
    slab[(int64_t)(p.chunk0 + ch.idx) * 4 + 0] = s0;
    slab[(int64_t)(p.chunk0 + ch.idx) * 4 + 1] = s1;
  }
}

// Diagonal grafting update before lr / clipping (DS:3502-3528).
__device__ inline float diag_graft(float g, float gden, float diag, int graft, float w1,
                                   float w2, float eps, float& new_diag) {
  const float scaled = normalized(graft) ? g / gden : g;  // grad / (||grad|| + 1e-25)
  if (graft == G_ADAGRAD || graft == G_ADAGRAD_N) new_diag = diag + scaled * scaled;
  else new_diag = w1 * diag + w2 * (scaled * scaled);
  return scaled / (sqrtf(new_diag) + eps);
}

// pass B (Adagrad/RMSProp only): slab[.][2] = sum u^2 of the unclipped update
__global__ __launch_bounds__(256) void transform_pass_b(const TParam* params, int count,
                                                        float* slab, ps_transform_config cfg) {
  __shared__ float red[4];
  const TChunk ch = find_chunk(params, count, blockIdx.x);
  const TParam p = params[ch.param];
  float gscale = 1.f;
  if (normalized(cfg.graft_type)) gscale = sqrtf(param_sum(slab, 0, p, red)) + T_EPS;
  const int64_t lo = (int64_t)ch.idx * TCHUNK;
  const int64_t hi = lo + TCHUNK < p.numel ? lo + TCHUNK : p.numel;
  float s = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    float nd;
    const float u = diag_graft(gload1(p.grad + i), gscale, gload1(p.diag_in + i), cfg.graft_type,
                               cfg.beta2_w1, cfg.beta2_w2, cfg.diagonal_epsilon, nd);
    s += u * u;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) slab[(int64_t)(p.chunk0 + ch.idx) * 4 + 2] = s;
}

// pass C: everything else, elementwise
__global__ __launch_bounds__(256) void transform_pass_c(const TParam* params, int count,
                                                        const float* slab, ps_transform_config cfg) {
  __shared__ float red[4];
  const TChunk ch = find_chunk(params, count, blockIdx.x);
  const TParam p = params[ch.param];
  const int graft = cfg.graft_type;
  const float r0 = param_sum(slab, 0, p, red);
  const float r1 = param_sum(slab, 1, p, red);
  float gscale = 1.f, clip_div = 1.f, gnorm;
  const float pm = cfg.decoupled_learning_rate ? 1.f : cfg.lr;  // DS:3549
  if (has_diag(graft)) {
    if (normalized(graft)) gscale = sqrtf(r0) + T_EPS;
    const float unorm = sqrtf(param_sum(slab, 2, p, red));
    if ((graft == G_RMSPROP || graft == G_RMSPROP_N) && cfg.clip_by_scaled_gradient_norm > 0.f) {
      const float scaled_norm = unorm / sqrtf((float)p.numel);  // DS:3531-3535
      clip_div = fmaxf(1.f, scaled_norm / cfg.clip_by_scaled_gradient_norm);
    }
    gnorm = unorm / clip_div * pm;
  } else {
    gnorm = sqrtf(r0) * pm;  // ||grad|| or sqrt(#nonzero) for sqrt-n
  }
  const float pnorm = p.pgrad ? sqrtf(r1) : gnorm;  // skipped: precond_grad = grafting_update
  const float mult = graft != G_NONE ? gnorm / (pnorm + T_EPS) : 1.f;  // DS:3566-3569
  const float w = cfg.moving_average_for_momentum ? 1.f - cfg.beta1 : 1.f;
  const float run = cfg.run_shampoo ? 1.f : 0.f;
  const bool coupled_wd = cfg.weight_decay != 0.f && !cfg.decoupled_weight_decay;
  const bool dec_wd = cfg.weight_decay != 0.f && cfg.decoupled_weight_decay;
  const float wd_lr = cfg.decoupled_learning_rate ? 1.f : cfg.lr;
  const float mm = cfg.decoupled_learning_rate ? cfg.lr : 1.f;  // DS:3610

  const int64_t lo = (int64_t)ch.idx * TCHUNK;
  const int64_t hi = lo + TCHUNK < p.numel ? lo + TCHUNK : p.numel;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const float g = gload1(p.grad + i);
    float gu;
    if (has_diag(graft)) {
      float nd;
      gu = diag_graft(g, gscale, gload1(p.diag_in + i), graft, cfg.beta2_w1, cfg.beta2_w2,
                      cfg.diagonal_epsilon, nd);
      gu = gu / clip_div;
      gstore1(p.diag_out + i, nd);
    } else if (graft == G_SQRT_N) {
      gu = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);
    } else {
      gu = g;
    }
    gu = gu * pm;
    const float pg = p.pgrad ? gload1(p.pgrad + i) : gu;
    float su = pg * mult;  // shampoo_update
    float sw = su, gw = gu;
    const float prm = (coupled_wd || dec_wd) ? gload1(p.param + i) : 0.f;
    if (coupled_wd) { sw = su + cfg.weight_decay * prm; gw = gu + cfg.weight_decay * prm; }
    const float smom = gload1(p.mom_in + i) * cfg.beta1 + w * sw;    // DS:3581-3586
    const float gmom = gload1(p.dmom_in + i) * cfg.beta1 + w * gw;
    const float mom_u = run * smom + (1.f - run) * gmom;
    const float wd_u = run * sw + (1.f - run) * gw;
    float nest = mom_u;
    if (cfg.nesterov) nest = w * wd_u + cfg.beta1 * mom_u;           // DS:3601-3602
    if (dec_wd) nest = nest + wd_lr * cfg.weight_decay * prm;
    gstore1(p.upd_out + i, -1.f * mm * nest);
    gstore1(p.mom_out + i, smom);
    gstore1(p.dmom_out + i, gmom);
  }
}

}  // namespace psk

using namespace psk;

extern "C" size_t ps_transform_grads_workspace_bytes(const ps_transform_desc* desc, int count) {
  if (!desc || count <= 0) return 0;
  size_t chunks = 0;
  for (int i = 0; i < count; ++i) chunks += (size_t)((desc[i].numel + TCHUNK - 1) / TCHUNK);
  return psh::align_up(sizeof(TParam) * count, 256) +
         psh::align_up(sizeof(float) * 4 * chunks, 256) + 1024;
}

extern "C" int ps_transform_grads_f32(void* stream, const ps_transform_desc* desc, int count,
                                      const ps_transform_config* cfg, void* workspace,
                                      size_t workspace_bytes) {
  PS_DEVICE_CHECK();
  if (!desc || !cfg || count <= 0 || !workspace) return PS_EINVAL;
  if (workspace_bytes < ps_transform_grads_workspace_bytes(desc, count)) return PS_EWORKSPACE;
  if (cfg->graft_type < 0 || cfg->graft_type > 6) return PS_EINVAL;
  const bool diag = cfg->graft_type == G_ADAGRAD || cfg->graft_type == G_RMSPROP ||
                    cfg->graft_type == G_RMSPROP_N || cfg->graft_type == G_ADAGRAD_N;
  std::vector<TParam> hp(count);
  long long chunk0 = 0;
  for (int i = 0; i < count; ++i) {
    const ps_transform_desc& d = desc[i];
    if (!d.grad || !d.mom_in || !d.mom_out || !d.dmom_in || !d.dmom_out || !d.upd_out ||
        d.numel < 0 || (diag && (!d.diag_in || !d.diag_out)) ||
        (cfg->weight_decay != 0.f && !d.param))
      return PS_EINVAL;
    TParam& p = hp[i];
    p.grad = d.grad; p.pgrad = d.pgrad; p.param = d.param; p.diag_in = d.diag_in;
    p.diag_out = d.diag_out; p.mom_in = d.mom_in; p.mom_out = d.mom_out;
    p.dmom_in = d.dmom_in; p.dmom_out = d.dmom_out; p.upd_out = d.upd_out;
    p.numel = d.numel;
    p.chunk0 = (int)chunk0;
    p.nchunks = (int)((d.numel + TCHUNK - 1) / TCHUNK);
    chunk0 += p.nchunks;
    if (chunk0 > 0x7fffffffLL) return PS_EUNSUPPORTED;
  }
  if (chunk0 == 0) return PS_OK;
  hipStream_t st = (hipStream_t)stream;
  psh::Arena ar(workspace, workspace_bytes);
  TParam* dp = ar.take<TParam>(count);
  float* slab = ar.take<float>(4 * (size_t)chunk0);
  if (ar.overflow) return PS_EWORKSPACE;
  PS_RC(psh::upload_async(st, dp, hp.data(), sizeof(TParam) * count));
  const dim3 grid((unsigned)chunk0), blk(256);
  hipLaunchKernelGGL(transform_pass_a, grid, blk, 0, st, dp, count, slab, *cfg);
  if (diag) hipLaunchKernelGGL(transform_pass_b, grid, blk, 0, st, dp, count, slab, *cfg);
  hipLaunchKernelGGL(transform_pass_c, grid, blk, 0, st, dp, count, slab, *cfg);
  PS_LAUNCH_CHECK();
  return PS_OK;
}
