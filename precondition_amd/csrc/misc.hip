// misc.hip — version/error strings, the seeded v0 of power_iteration, a plain
// batched product and mat_power (reference: DS:642-643, DS:655-678).
#include <string.h>

#include <vector>

#include <new>

#include "common.h"
#include "gemm_core.hip.h"

using namespace psk;

extern "C" int ps_version(void) { return PS_VERSION; }

extern "C" const char* ps_error_string(int code) {
  switch (code) {
    case PS_OK: return "ok";
    case PS_EINVAL: return "invalid argument";
    case PS_EWORKSPACE: return "workspace too small";
    case PS_EUNSUPPORTED: return "unsupported size or exponent";
    case PS_EINTERNAL: return "internal error";
    case PS_EDEVICE: return "wrong HIP device current (the library serves one device per process)";
    case PS_ECOMM: return "RCCL unavailable or an RCCL call failed (ps_comm_last_error)";
    default:
      return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

// numpy.random.RandomState(1729).uniform(-1.0, 1.0, n).astype(float32):
// MT19937 seeded with init_genrand(1729); each double is built from two 32-bit
// draws, (a>>5, b>>6) -> (a*2^26 + b) / 2^53; uniform = low + (high-low)*u.
extern "C" int ps_power_iteration_v0(int n, float* out) {
  if (n < 0 || (n > 0 && !out)) return PS_EINVAL;
  uint32_t mt[624];
  mt[0] = 1729u;
  for (int i = 1; i < 624; ++i)
    mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  int pos = 624;
  auto next = [&]() -> uint32_t {
    if (pos >= 624) {
      for (int k = 0; k < 624; ++k) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      pos = 0;
    }
    uint32_t y = mt[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  };
  for (int i = 0; i < n; ++i) {
    const uint32_t a = next() >> 5, b = next() >> 6;
    const double u = (a * 67108864.0 + b) / 9007199254740992.0;
    out[i] = (float)(-1.0 + 2.0 * u);
  }
  return PS_OK;
}

namespace psk {

constexpr int GBK = 16;

// C[b] = op(A[b]) * op(B[b]), row-major, any sizes / alignment (guarded loads).
template <int LA, int LB>
__global__ __launch_bounds__(256, 2) void gemm_kernel(
    const float* a, const float* b, float* c, int M, int N, int K, int lda, int ldb,
    int ldc, int64_t sa, int64_t sb, int64_t sc, int tiles_m, int tiles_n, int veca,
    int vecb) {
  __shared__ __align__(16) float smem[SmemCfg<GBK>::TOTAL];
  const int per = tiles_m * tiles_n;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int bi = t / per, tm = (t % per) / tiles_n, tn = t % tiles_n;
  Operand A{a + bi * sa, lda, tm * TILE, M, K, veca != 0};
  Operand B{b + bi * sb, ldb, tn * TILE, N, K, vecb != 0};
  f32x16 acc[2][2];
  gemm_tile<LA, LB, GBK, true>(A, B, K, smem, acc);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  float* C = c + bi * sc;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tm * TILE + acc_row(wm, i, r, lane);
        const int col = tn * TILE + acc_col(wn, j, lane);
        if (row < M && col < N) gstore1(C + (int64_t)row * ldc + col, acc[i][j][r]);
      }
}

}  // namespace psk

static bool vec_ok(const float* p, int ld, int64_t stride) {
  return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0) && (stride % 4 == 0);
}

extern "C" int ps_gemm_f32(void* stream, int transa, int transb, const float* a,
                           const float* b, float* c, int m, int n, int k, int lda,
                           int ldb, int ldc, int batch, int64_t stride_a,
                           int64_t stride_b, int64_t stride_c) {
  PS_DEVICE_CHECK();
  if (!a || !b || !c || m < 1 || n < 1 || k < 1 || batch < 1 || ldc < n ||
      lda < (transa ? m : k) || ldb < (transb ? k : n))
    return PS_EINVAL;
  const int tm = (m + TILE - 1) / TILE, tn = (n + TILE - 1) / TILE;
  const int64_t grid = (int64_t)tm * tn * batch;
  if (grid > 0x7fffffff) return PS_EUNSUPPORTED;
  const int va = vec_ok(a, lda, stride_a) ? 1 : 0, vb = vec_ok(b, ldb, stride_b) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g((unsigned)grid), blk(256);
  // A: rows are mn. transa=0 -> k-contiguous (KC); transa=1 -> mn-contiguous (MC).
  // B: cols are mn. transb=0 -> mn-contiguous (MC); transb=1 -> k-contiguous (KC).
#define PS_GEMM_LAUNCH(LA, LB)                                                        \
  hipLaunchKernelGGL((gemm_kernel<LA, LB>), g, blk, 0, st, a, b, c, m, n, k, lda, ldb, \
                     ldc, stride_a, stride_b, stride_c, tm, tn, va, vb)
  if (!transa && !transb) PS_GEMM_LAUNCH(KC, MC);
  else if (!transa && transb) PS_GEMM_LAUNCH(KC, KC);
  else if (transa && !transb) PS_GEMM_LAUNCH(MC, MC);
  else PS_GEMM_LAUNCH(MC, KC);
#undef PS_GEMM_LAUNCH
  PS_LAUNCH_CHECK();
  return PS_OK;
}

namespace psk {
struct GTask {
  const float* a; const float* b; float* c;
  int m, n, k, lda, ldb, ldc, veca, vecb;
  int ksplit, kchunk;   // split-K: tile (tm, tn, ks) covers k in [ks*kchunk, (ks+1)*kchunk)
  float* partial;       // [ksplit][m][n] when ksplit > 1
};
struct GTile { int task; short tm, tn; int ks; };

// One-row products (m = 1: the preconditioner applied to a vector block, g^T P) are a
// matrix-vector pass: a 128-column strip of B is streamed once by the VALU (16-byte loads,
// 8 in flight per lane, rows dealt to 8 lane groups and combined through LDS in a fixed
// order) instead of running 128 x 128 MFMA tiles with one useful row.  The ViT-B tree has 149
// such blocks: ~1000 of the 6300 tiles of an application stage.
__device__ inline void gemv_row_tile(const float* a, int64_t astride, const float* b, int ldb,
                                     float* c, int n, int k, int col0, bool vecb, float* smem) {
  const int tid = threadIdx.x;
  const int cg = tid & 31, rg = tid >> 5;
  const int col = col0 + cg * 4;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (vecb && col + 3 < n) {
    int kk = rg;
    for (; kk + 56 < k; kk += 64) {
      f32x4 v[8];
      float av[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = gload4(b + (int64_t)(kk + 8 * u) * ldb + col);
        av[u] = gload1(a + (int64_t)(kk + 8 * u) * astride);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = __fmaf_rn(av[u], v[u][e], s[e]);
    }
    for (; kk < k; kk += 8) {
      const f32x4 v = gload4(b + (int64_t)kk * ldb + col);
      const float av = gload1(a + (int64_t)kk * astride);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = __fmaf_rn(av, v[e], s[e]);
    }
  } else {
    for (int kk = rg; kk < k; kk += 8) {
      const float av = gload1(a + (int64_t)kk * astride);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < n) s[e] = __fmaf_rn(av, gload1(b + (int64_t)kk * ldb + col + e), s[e]);
    }
  }
  float* red = smem;  // [8][128]
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rg * 128 + cg * 4 + e] = s[e];
  __syncthreads();
  if (tid < 128 && col0 + tid < n) {
    float t = red[tid];
#pragma unroll
    for (int r = 1; r < 8; ++r) t = __fadd_rn(t, red[r * 128 + tid]);
    gstore1(c + col0 + tid, t);
  }
}

constexpr int GBK_FAST = 32;  // interior tiles: unguarded loads, two register sets in flight
constexpr int GEMM_GROUPED_LDS_BYTES = SmemCfg<GBK_FAST>::TOTAL * (int)sizeof(float);

// Interior tiles of 16-byte aligned operands whose k range is a whole number of 32-deep
// K-tiles run the unguarded, two-register-set K loop of the Newton / statistics kernels and
// an epilogue with wave-uniform addressing; edge tiles and unaligned operands the guarded
// BK = 16 loop.  Same products in the same k order: bit-identical results.
template <int LA, int LB>
__global__ __launch_bounds__(256, 2) void gemm_grouped_kernel(const GTask* tasks,
                                                              const GTile* tiles, int ntiles) {
  extern __shared__ __align__(16) float smem[];
  const GTile te = tiles[blockIdx.x];   // dispatch order (common.h deal_to_xcds)
  if (te.task < 0) return;
  const GTask tk = tasks[te.task];
  const int k0 = te.ks * tk.kchunk;
  const int kext = min(tk.kchunk, tk.k - k0);
  // element (mn, k) of a KC operand lies at p[mn*ld + k], of an MC operand at p[k*ld + mn]
  const float* ap = tk.a + (LA == KC ? (int64_t)k0 : (int64_t)k0 * tk.lda);
  const float* bp = tk.b + (LB == KC ? (int64_t)k0 : (int64_t)k0 * tk.ldb);
  Operand A{ap, tk.lda, te.tm * TILE, tk.m, kext, tk.veca != 0};
  Operand B{bp, tk.ldb, te.tn * TILE, tk.n, kext, tk.vecb != 0};
  f32x16 acc[2][2];
  float* out = tk.ksplit > 1 ? tk.partial + (int64_t)te.ks * tk.m * tk.n : tk.c;
  const int ldo = tk.ksplit > 1 ? tk.n : tk.ldc;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  if (LB == MC && tk.m == 1 && tk.ksplit == 1) {
    gemv_row_tile(tk.a, LA == KC ? 1 : tk.lda, tk.b, tk.ldb, tk.c, tk.n, tk.k, te.tn * TILE,
                  tk.vecb != 0, smem);
    return;
  }
  const bool fast = tk.veca != 0 && tk.vecb != 0 && kext % GBK_FAST == 0 &&
                    (te.tm + 1) * TILE <= tk.m && (te.tn + 1) * TILE <= tk.n;
  if (fast) {
    gemm_tile<LA, LB, GBK_FAST, false, true>(A, B, kext, smem, acc);
    const int lane_off = 4 * (lane >> 5) * ldo + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float* blk = out + (int64_t)(te.tm * TILE + wm * 64 + i * 32) * ldo + te.tn * TILE +
                     wn * 64 + j * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          gstore1(blk + (int64_t)((r & 3) + 8 * (r >> 2)) * ldo + lane_off, acc[i][j][r]);
      }
    return;
  }
  gemm_tile<LA, LB, GBK, true>(A, B, kext, smem, acc);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = te.tm * TILE + acc_row(wm, i, r, lane);
        const int col = te.tn * TILE + acc_col(wn, j, lane);
        if (row < tk.m && col < tk.n) gstore1(out + (int64_t)row * ldo + col, acc[i][j][r]);
      }
}

// c = sum over the K splits of the partial products, in split order (deterministic).
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const GTask* tasks,
                                                                 const int* task_ids) {
  const GTask tk = tasks[task_ids[blockIdx.x]];
  const int64_t mn = (int64_t)tk.m * tk.n;
  for (int64_t e = blockIdx.y * 256 + threadIdx.x; e < mn; e += (int64_t)gridDim.y * 256) {
    // all partials of the element in flight, then summed in split order (deterministic)
    float v = 0.f;
    int s = 0;
    for (; s + 4 <= tk.ksplit; s += 4) {
      const float p0 = tk.partial[(s + 0) * mn + e], p1 = tk.partial[(s + 1) * mn + e],
                  p2 = tk.partial[(s + 2) * mn + e], p3 = tk.partial[(s + 3) * mn + e];
      v = (((v + p0) + p1) + p2) + p3;
    }
    for (; s < tk.ksplit; ++s) v += tk.partial[s * mn + e];
    tk.c[(e / tk.n) * tk.ldc + e % tk.n] = v;
  }
}
}  // namespace psk

// Split-K plan of one grouped call: products whose output has far fewer tiles than the GPU
// has CUs but a long contraction (the b x n @ n x b Gram / Rayleigh-Ritz products of the
// subspace iteration: ONE 128 x 128 tile with k = 4096) are cut along k into `ksplit`
// partial products, each in its own workgroup, summed by gemm_splitk_reduce_kernel.
// Calls with enough tiles (the per-step application of the preconditioners) are unchanged.
static int splitk_for(int total_tiles, int k) {
  // one workgroup per CU with a long contraction (the float32 C @ X of the Rayleigh-Ritz step of the
  // FD branch: 256 tiles, k = 4096) leaves the CU waiting on every K-tile: two per CU from k = 2048
  if (total_tiles >= 512 || k < 512 || (total_tiles >= 256 && k < 2048)) return 1;
  int s = (512 + total_tiles - 1) / total_tiles;   // aim at ~2 workgroups per CU
  s = std::min(s, k / 256);                        // at least 256 k per split
  return std::max(1, std::min(s, 64));
}

static size_t grouped_total_tiles(const ps_gemm_desc* d, int count) {
  size_t tiles = 0;
  for (int i = 0; i < count; ++i)
    tiles += (size_t)((d[i].m + TILE - 1) / TILE) * ((d[i].n + TILE - 1) / TILE);
  return tiles;
}

static size_t grouped_gemm_bytes(const ps_gemm_desc* d, int count) {
  const size_t tiles = grouped_total_tiles(d, count);
  size_t split_tiles = 0, partial = 0;
  for (int i = 0; i < count; ++i) {
    const int s = splitk_for((int)std::min<size_t>(tiles, 1 << 30), d[i].k);
    split_tiles += (size_t)s * ((d[i].m + TILE - 1) / TILE) * ((d[i].n + TILE - 1) / TILE);
    if (s > 1) partial += psh::align_up((size_t)s * d[i].m * d[i].n * sizeof(float), 256) + 256;
  }
  return 4 * (psh::align_up(sizeof(GTask) * count, 256) + 256) +
         4 * (psh::align_up(sizeof(GTile) * split_tiles * psh::NXCD, 256) + 256) +
         psh::align_up(sizeof(int) * count, 256) + 256 + partial + 1024;
}

extern "C" size_t ps_gemm_grouped_workspace_bytes(const ps_gemm_desc* desc, int count) {
  if (!desc || count <= 0) return 0;
  return grouped_gemm_bytes(desc, count);
}

// A grouped product as a reusable plan (ps_gemm_grouped_plan_*): the task / tile tables are
// built and uploaded once and the launch is two kernel launches with no host work -- the
// subspace iteration of the FD branch repeats the same ~9 products on the same buffers every
// round, and building their tables from Python (~90 us per product) was 4 of the 10 ms of a
// one-factor update.
struct ps_gemm_plan {
  GTask* dt[4] = {nullptr, nullptr, nullptr, nullptr};
  GTile* dl[4] = {nullptr, nullptr, nullptr, nullptr};
  int* di[4] = {nullptr, nullptr, nullptr, nullptr};
  int ntiles[4] = {0, 0, 0, 0}, nsplit[4] = {0, 0, 0, 0};
};

static int gplan_build(hipStream_t st, const ps_gemm_desc* desc, int count, void* workspace,
                       size_t workspace_bytes, ps_gemm_plan& pl) {
  if (!desc || count <= 0 || !workspace) return PS_EINVAL;
  if (workspace_bytes < grouped_gemm_bytes(desc, count)) return PS_EWORKSPACE;
  std::vector<GTask> tasks[4];
  std::vector<GTile> tiles[4];
  std::vector<int> split_ids[4];
  psh::Arena ar(workspace, workspace_bytes);
  const int total_tiles = (int)std::min<size_t>(grouped_total_tiles(desc, count), 1 << 30);
  for (int i = 0; i < count; ++i) {
    const ps_gemm_desc& d = desc[i];
    if (!d.a || !d.b || !d.c || d.m < 1 || d.n < 1 || d.k < 1 || d.ldc < d.n ||
        d.lda < (d.transa ? d.m : d.k) || d.ldb < (d.transb ? d.k : d.n) ||
        d.lda > 0x7fffffff || d.ldb > 0x7fffffff || d.ldc > 0x7fffffff)
      return PS_EINVAL;
    const int g = (d.transa ? 2 : 0) + (d.transb ? 1 : 0);
    GTask t{d.a, d.b, d.c, d.m, d.n, d.k, (int)d.lda, (int)d.ldb, (int)d.ldc,
            vec_ok(d.a, (int)d.lda, 0) ? 1 : 0, vec_ok(d.b, (int)d.ldb, 0) ? 1 : 0, 1, d.k,
            nullptr};
    const int s = splitk_for(total_tiles, d.k);
    if (s > 1) {
      t.kchunk = psh::round_up((d.k + s - 1) / s, 128);   // keeps 16-byte load alignment
      t.ksplit = (d.k + t.kchunk - 1) / t.kchunk;
      if (t.ksplit > 1) t.partial = ar.take<float>((size_t)t.ksplit * d.m * d.n);
      else t.kchunk = d.k;
    }
    const int id = (int)tasks[g].size();
    tasks[g].push_back(t);
    if (t.ksplit > 1) split_ids[g].push_back(id);
  }
  // tile lists in dispatch order: tasks dealt to the XCDs by cost (K-tiles per tile + an
  // epilogue term; the mat-vec tiles of one-row products are short streams)
  for (int g = 0; g < 4; ++g) {
    if (tasks[g].empty()) continue;
    std::vector<int> ntile(tasks[g].size());
    std::vector<int64_t> cost(tasks[g].size());
    for (size_t i = 0; i < tasks[g].size(); ++i) {
      const GTask& t = tasks[g][i];
      const int tm = (t.m + TILE - 1) / TILE, tn = (t.n + TILE - 1) / TILE;
      ntile[i] = tm * tn * t.ksplit;
      const bool gemv = (g == 0 || g == 2) && t.m == 1 && t.ksplit == 1;
      cost[i] = gemv ? 2 : (t.kchunk + 31) / 32 + 4;
    }
    std::vector<psh::DealUnit> units[psh::NXCD];
    psh::deal_to_xcds(ntile, cost, units);
    std::vector<GTile> lists[psh::NXCD];
    for (int x = 0; x < psh::NXCD; ++x)
      for (const psh::DealUnit& un : units[x]) {
        const GTask& t = tasks[g][un.task];
        const int tn = (t.n + TILE - 1) / TILE;
        for (int idx = un.first; idx < un.first + un.count; ++idx) {
          const int ks = idx % t.ksplit, ab = idx / t.ksplit;
          lists[x].push_back({un.task, (short)(ab / tn), (short)(ab % tn), ks});
        }
      }
    psh::interleave_xcd_lists(lists, GTile{-1, 0, 0, 0}, tiles[g]);
  }
  for (int g = 0; g < 4; ++g) {
    pl.dt[g] = ar.take<GTask>(std::max<size_t>(tasks[g].size(), 1));
    pl.dl[g] = ar.take<GTile>(std::max<size_t>(tiles[g].size(), 1));
    pl.di[g] = ar.take<int>(std::max<size_t>(split_ids[g].size(), 1));
    pl.ntiles[g] = (int)tiles[g].size();
    pl.nsplit[g] = (int)split_ids[g].size();
  }
  if (ar.overflow) return PS_EWORKSPACE;
  for (int g = 0; g < 4; ++g) {
    if (tasks[g].empty()) continue;
    PS_RC(psh::upload_async(st, pl.dt[g], tasks[g].data(), sizeof(GTask) * tasks[g].size()));
    PS_RC(psh::upload_async(st, pl.dl[g], tiles[g].data(), sizeof(GTile) * tiles[g].size()));
    PS_RC(psh::upload_async(st, pl.di[g], split_ids[g].data(), sizeof(int) * split_ids[g].size()));
  }
  static const bool lds_ok = [] {
    const void* ks[4] = {(const void*)gemm_grouped_kernel<KC, MC>, (const void*)gemm_grouped_kernel<KC, KC>,
                         (const void*)gemm_grouped_kernel<MC, MC>, (const void*)gemm_grouped_kernel<MC, KC>};
    for (const void* k : ks)
      (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                GEMM_GROUPED_LDS_BYTES);
    return true;
  }();
  (void)lds_ok;
  return PS_OK;
}

static int gplan_launch(hipStream_t st, const ps_gemm_plan& pl) {
  const dim3 blk(256);
#define PS_GG(G, LA, LB)                                                                   \
  if (pl.ntiles[G] > 0) {                                                                  \
    hipLaunchKernelGGL((gemm_grouped_kernel<LA, LB>), dim3((unsigned)pl.ntiles[G]), blk,   \
                       GEMM_GROUPED_LDS_BYTES, st, pl.dt[G], pl.dl[G], pl.ntiles[G]);      \
    if (pl.nsplit[G] > 0)                                                                  \
      hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)pl.nsplit[G], 256), blk, 0, \
                         st, pl.dt[G], pl.di[G]);                                          \
  }
  PS_GG(0, KC, MC);
  PS_GG(1, KC, KC);
  PS_GG(2, MC, MC);
  PS_GG(3, MC, KC);
#undef PS_GG
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_gemm_grouped_f32(void* stream, const ps_gemm_desc* desc, int count,
                                   void* workspace, size_t workspace_bytes) {
  PS_DEVICE_CHECK();
  ps_gemm_plan pl;
  PS_RC(gplan_build((hipStream_t)stream, desc, count, workspace, workspace_bytes, pl));
  return gplan_launch((hipStream_t)stream, pl);
}

extern "C" int ps_gemm_grouped_plan_create(void* stream, const ps_gemm_desc* desc, int count,
                                           void* workspace, size_t workspace_bytes,
                                           ps_gemm_plan** plan) {
  PS_DEVICE_CHECK();
  if (!plan) return PS_EINVAL;
  *plan = nullptr;
  ps_gemm_plan* pl = new (std::nothrow) ps_gemm_plan();
  if (!pl) return PS_EINTERNAL;
  const int rc = gplan_build((hipStream_t)stream, desc, count, workspace, workspace_bytes, *pl);
  if (rc != 0) { delete pl; return rc; }
  *plan = pl;
  return PS_OK;
}

extern "C" int ps_gemm_grouped_plan_launch(void* stream, const ps_gemm_plan* plan) {
  PS_DEVICE_CHECK();
  if (!plan) return PS_EINVAL;
  return gplan_launch((hipStream_t)stream, *plan);
}

extern "C" int ps_gemm_grouped_plan_destroy(ps_gemm_plan* plan) {
  delete plan;
  return PS_OK;
}

static int ps_gemm_nn_f32(void* stream, const float* a, const float* b, float* c, int m,
                          int n, int k, int lda, int ldb, int ldc, int batch,
                          int64_t sa, int64_t sb, int64_t sc) {
  return ps_gemm_f32(stream, 0, 0, a, b, c, m, n, k, lda, ldb, ldc, batch, sa, sb, sc);
}

// mat_power: the reference's loop (DS:663-677) with the exact "@ I" skipped.
extern "C" size_t ps_mat_power_workspace_bytes(int n, int p) {
  if (n < 1 || p < 0) return 0;
  int bits = 0;
  for (int i = p; i > 0; i >>= 1) ++bits;
  return (size_t)(2 * bits + 2) * n * n * sizeof(float) + 1024;
}

extern "C" int ps_mat_power_f32(void* stream, const float* m, int n, int ldm, int p,
                                float* out, int ldo, void* workspace,
                                size_t workspace_bytes) {
  PS_DEVICE_CHECK();
  if (!m || !out || n < 1 || p < 0 || ldm < n || ldo < n) return PS_EINVAL;
  if (workspace_bytes < ps_mat_power_workspace_bytes(n, p) || !workspace)
    return PS_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  psh::Arena ar(workspace, workspace_bytes);
  const size_t sq = (size_t)n * n;
  const float* mat = m;
  int ldmat = ldm;
  const float* power = nullptr;
  int ldpow = n;
  int i = p;
  while (i > 0) {
    if (i & 1) {
      if (!power) { power = mat; ldpow = ldmat; }
      else {
        float* t = ar.take<float>(sq);
        int rc = ps_gemm_nn_f32(stream, mat, power, t, n, n, n, ldmat, ldpow, n, 1, 0, 0, 0);
        if (rc) return rc;
        power = t; ldpow = n;
      }
    }
    i >>= 1;
    if (i > 0) {
      float* t = ar.take<float>(sq);
      int rc = ps_gemm_nn_f32(stream, mat, mat, t, n, n, n, ldmat, ldmat, n, 1, 0, 0, 0);
      if (rc) return rc;
      mat = t; ldmat = n;
    }
  }
  if (!power) {  // p == 0: identity
    std::vector<float> eye(sq, 0.f);
    for (int r = 0; r < n; ++r) eye[(size_t)r * n + r] = 1.f;
    PS_HIP(hipMemcpy2DAsync(out, (size_t)ldo * 4, eye.data(), (size_t)n * 4, (size_t)n * 4,
                            n, hipMemcpyHostToDevice, st));
    PS_HIP(hipStreamSynchronize(st));
    return PS_OK;
  }
  PS_HIP(hipMemcpy2DAsync(out, (size_t)ldo * 4, power, (size_t)ldpow * 4, (size_t)n * 4, n,
                          hipMemcpyDeviceToDevice, st));
  return PS_OK;
}
