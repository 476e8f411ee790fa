// Dev (round 6): is the Newton product kernel's loss the BARRIER COUPLING of its four wavefronts?
//
// The stage kernel computes a 128 x 128 tile per workgroup: four wavefronts (one per SIMD) share the LDS image of
// every K-tile and meet at a workgroup barrier once per K-tile; two workgroups share a CU, so every wavefront
// shares its SIMD's MFMA pipe with a wavefront of the OTHER workgroup, which is in another phase -- the four
// siblings progress unevenly and the barrier makes each wait for the slowest, sixteen times per tile.
// Variant measured here: WAVE-PRIVATE tiles -- every wavefront computes a 64 x 64 tile on its own (own 18 KB LDS image,
// own global loads: twice the L2 -> LDS traffic per flop), eight independent wavefronts per CU, NO barrier anywhere.
// Same MFMA (v_mfma_f32_32x32x2_f32), same fragment layout and k order as gemm_core.hip.h (bit-identical sums).
// Both kernels compute C_b = A_b B_b^T for a batch of n x n float32 matrices (both operands k-contiguous, the
// layout of the symmetric Newton products) with a plain store epilogue.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I precondition_amd/csrc tools/bench_wavetile.hip -o tools/bin/bench_wavetile
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "gemm_core.hip.h"

using namespace psk;

// ---- reference structure: one 128 x 128 tile per workgroup (gemm_core.hip.h, the stage kernel's K loop) -----
// SEG: segmented accumulation (deep_run_pipe_seg: one register set, totals every 128 k), as the Newton products run;
// TRI: only the tiles tm <= tn of every matrix (T (T + 1) / 2 per matrix) and each off-diagonal tile also stored
//      transposed (store_tile_transposed_v4): the symmetric products of the Newton stage kernel;
// EXTRA: a second output per element (alpha * v, direct and mirrored): the Mi of an M update.
template <bool PIPE, bool SEG = false, bool TRI = false, bool EXTRA = false>
__global__ __launch_bounds__(256, 2) void wg_tile_kernel(const float* A, const float* B, float* C, int n, int T,
                                                        float* C2 = nullptr) {
  extern __shared__ __align__(16) float smem[];
  const int tiles = TRI ? T * (T + 1) / 2 : T * T;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mat = bid / tiles, t = bid % tiles;
  int tm, tn;
  if (TRI) {   // row-major upper triangle: (0,0) (0,1) .. (0,T-1) (1,1) ..
    int r = 0, left = t;
    while (left >= T - r) { left -= T - r; ++r; }
    tm = r; tn = r + left;
  } else {
    tm = t / T; tn = t % T;
  }
  Operand a{A + (size_t)mat * n * n, n, tm * TILE, n, n, true};
  Operand b{B + (size_t)mat * n * n, n, tn * TILE, n, n, true};
  f32x16 acc[2][2];
  gemm_tile<KC, KC, 32, false, true, PIPE, SEG>(a, b, n, smem, acc, nullptr, SEG);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1;
  float* c = C + (size_t)mat * n * n;
  float* c2 = EXTRA ? C2 + (size_t)mat * n * n : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const size_t o = (size_t)(tm * TILE + acc_row(wm, i, r, lane)) * n + tn * TILE + acc_col(wn, j, lane);
        c[o] = acc[i][j][r];
        if (EXTRA) c2[o] = -0.25f * acc[i][j][r];
      }
  if (TRI && tm != tn) store_tile_transposed_v4<false>(acc, smem, c, c2, -0.25f, n, tn * TILE, tm * TILE);
}

// The same "Newton square" with the stage kernel's addressing: tile entry -> block descriptor (pointers by buffer id,
// sizes) + block state (phase, current buffer) -> operands: three dependent global loads before the first tile load.
struct MbBlock { const float* buf[4]; float* out; int n, npad; short pa, pb; int pad_[3]; };
struct MbState { int phase, cur, it, general; float err; int pad_[3]; };
struct MbTile { int block; short tm, tn; };
__global__ __launch_bounds__(256, 2) void wg_tile_indirect_kernel(const MbBlock* blocks, const MbState* states,
                                                                 const MbTile* tiles, int ntiles) {
  extern __shared__ __align__(16) float smem[];
  const MbTile te = tiles[xcd_remap(blockIdx.x, ntiles)];
  const MbBlock* nb = &blocks[te.block];
  const MbState* st = &states[te.block];
  if (st->phase != 1 || (st->it == 0 && te.block < 0)) return;
  const int n = nb->n, ld = nb->npad, tm = te.tm, tn = te.tn;
  const float* pa = nb->buf[(nb->pa + st->cur) & 3];
  const float* pb = nb->buf[(nb->pb + st->cur) & 3];
  Operand a{pa, ld, tm * TILE, ld, ld, true};
  Operand b{pb, ld, tn * TILE, ld, ld, true};
  f32x16 acc[2][2];
  gemm_tile<KC, KC, 32, false, true, true, true>(a, b, n, smem, acc, nullptr, st->general == 0);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1;
  float* c = nb->out;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        c[(size_t)(tm * TILE + acc_row(wm, i, r, lane)) * ld + tn * TILE + acc_col(wn, j, lane)] = acc[i][j][r];
  if (tm != tn) store_tile_transposed_v4<false>(acc, smem, c, nullptr, 0.f, ld, tn * TILE, tm * TILE);
}

// ---- wave-private 64 x 64 tiles ------------------------------------------------------------------------------
constexpr int WBK = 32, WLD = WBK + 4, WROWS = 64;
constexpr int WAVE_LDS = 2 * WROWS * WLD;   // floats per wavefront (A image + B image)

struct WFrag { float a[2][4], b[2][4]; };

__device__ __forceinline__ void w_read(const float* sA, const float* sB, int c, int i, int h, WFrag& f) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x4 va = *reinterpret_cast<const f32x4*>(sA + (t * 32 + i) * WLD + 8 * c + 4 * h);
    const f32x4 vb = *reinterpret_cast<const f32x4*>(sB + (t * 32 + i) * WLD + 8 * c + 4 * h);
    f.a[t][0] = va[0]; f.a[t][1] = va[1]; f.a[t][2] = va[2]; f.a[t][3] = va[3];
    f.b[t][0] = vb[0]; f.b[t][1] = vb[1]; f.b[t][2] = vb[2]; f.b[t][3] = vb[3];
  }
}
__device__ __forceinline__ void w_mfma(const WFrag& f, f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[tm][s], f.b[tn][s], acc[tm][tn], 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void wave_tile_kernel(const float* A, const float* B, float* C, int n, int T,
                                                          int ntiles_total) {
  extern __shared__ __align__(16) float smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // four consecutive tiles of one workgroup share their A row panel (same tm): L2 / L1 reuse
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= ntiles_total) return;      // no barrier anywhere: a wavefront may leave early
  const int tiles = T * T, mat = tile / tiles, t = tile % tiles, tm = t / T, tn = t % T;
  float* sA = smem + wave * WAVE_LDS;
  float* sB = sA + WROWS * WLD;
  const float* Ab = A + (size_t)mat * n * n + (size_t)tm * 64 * n;
  const float* Bb = B + (size_t)mat * n * n + (size_t)tn * 64 * n;
  const int lr = lane >> 3, lq = (lane & 7) * 4;   // loader: row lr + 8 v, k = lq
  const int i = lane & 31, h = lane >> 5;
  f32x4 ra[8], rb[8];
  auto gl = [&](int k0) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      ra[v] = *(const f32x4 PS_GLOBAL*)(Ab + (size_t)(lr + 8 * v) * n + k0 + lq);
      rb[v] = *(const f32x4 PS_GLOBAL*)(Bb + (size_t)(lr + 8 * v) * n + k0 + lq);
    }
  };
  auto st = [&]() {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      *reinterpret_cast<f32x4*>(sA + (lr + 8 * v) * WLD + lq) = ra[v];
      *reinterpret_cast<f32x4*>(sB + (lr + 8 * v) * WLD + lq) = rb[v];
    }
  };
  f32x16 acc[2][2];
  zero_acc(acc);
  const int nk = n / WBK;
  gl(0);
  st();
  gl(nk > 1 ? WBK : 0);
  WFrag f0, f1;
  w_read(sA, sB, 0, i, h, f0);
  for (int kt = 0; kt < nk; ++kt) {
    // LDS holds K-tile kt (chunk 0 already in f0); the register set holds K-tile kt + 1
    PS_FENCE();
    w_read(sA, sB, 1, i, h, f1);
    PS_FENCE();
    w_mfma(f0, acc);
    PS_FENCE();
    w_read(sA, sB, 2, i, h, f0);
    PS_FENCE();
    w_mfma(f1, acc);
    PS_FENCE();
    w_read(sA, sB, 3, i, h, f1);
    PS_FENCE();
    w_mfma(f0, acc);
    PS_FENCE();
    // every read of this K-tile is issued; LDS operations of a wavefront execute in order, so the image may be
    // overwritten now.  Then request K-tile kt + 2 and read chunk 0 of K-tile kt + 1 under the last chunk's MFMAs.
    if (kt + 1 < nk) {
      st();
      gl(min(kt + 2, nk - 1) * WBK);
      PS_FENCE();
      w_read(sA, sB, 0, i, h, f0);
    }
    PS_FENCE();
    w_mfma(f1, acc);
  }
  float* c = C + (size_t)mat * n * n;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        c[(size_t)(tm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * n + tn * 64 + b * 32 + (lane & 31)] =
            acc[a][b][r];
}

int main() {
  for (auto cfg : std::vector<std::pair<int, int>>{{256, 512}, {64, 1024}}) {
    const int nb = cfg.first, n = cfg.second;
    const size_t el = (size_t)nb * n * n;
    float *A, *B, *C0, *C1;
    hipMalloc(&A, el * 4); hipMalloc(&B, el * 4); hipMalloc(&C0, el * 4); hipMalloc(&C1, el * 4);
    std::vector<float> h(el);
    srand(7);
    for (size_t k = 0; k < el; ++k) h[k] = (float)(rand() % 2001 - 1000) / 1000.f;
    hipMemcpy(A, h.data(), el * 4, hipMemcpyHostToDevice);
    for (size_t k = 0; k < el; ++k) h[k] = (float)(rand() % 2001 - 1000) / 1000.f;
    hipMemcpy(B, h.data(), el * 4, hipMemcpyHostToDevice);
    const double flops = 2.0 * nb * (double)n * n * n;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int T128 = n / 128, T64 = n / 64;
    const size_t lds128 = SmemCfg<32>::TOTAL * sizeof(float), lds64 = 4 * WAVE_LDS * sizeof(float);
    hipFuncSetAttribute((const void*)wave_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
    const int nt64 = nb * T64 * T64;
    float* C2; hipMalloc(&C2, el * 4);
    const void* fns[] = {(const void*)wg_tile_kernel<true>, (const void*)wg_tile_kernel<false>, (const void*)wg_tile_kernel<true, true>,
                         (const void*)wg_tile_kernel<true, false, true>, (const void*)wg_tile_kernel<true, true, true>,
                         (const void*)wg_tile_kernel<true, true, true, true>};
    for (const void* f : fns) hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
    const char* names[] = {"128x128 tile, pipelined K loop, full products", "128x128 tile, plain deep K loop, full products",
                           "  + segmented accumulation (one register set)", "  + upper tile triangle, mirrored stores (no segments)",
                           "  + segments + triangle + mirror (a Newton square)", "  + second output, direct and mirrored (an M update)",
                           "wave-private tiles 64x64, no barriers, full products"};
    const int tri = T128 * (T128 + 1) / 2, full = T128 * T128;
    for (int variant = 0; variant < 7; ++variant) {
      float best = 1e9f;
      const bool is_tri = variant >= 3 && variant <= 5;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) {
          const dim3 g(nb * (is_tri ? tri : full)), blk(256);
          switch (variant) {
            case 0: hipLaunchKernelGGL((wg_tile_kernel<true>), g, blk, lds128, 0, A, B, C0, n, T128, nullptr); break;
            case 1: hipLaunchKernelGGL((wg_tile_kernel<false>), g, blk, lds128, 0, A, B, C0, n, T128, nullptr); break;
            case 2: hipLaunchKernelGGL((wg_tile_kernel<true, true>), g, blk, lds128, 0, A, B, C0, n, T128, nullptr); break;
            case 3: hipLaunchKernelGGL((wg_tile_kernel<true, false, true>), g, blk, lds128, 0, A, A, C2, n, T128, nullptr); break;
            case 4: hipLaunchKernelGGL((wg_tile_kernel<true, true, true>), g, blk, lds128, 0, A, A, C2, n, T128, nullptr); break;
            case 5: hipLaunchKernelGGL((wg_tile_kernel<true, true, true, true>), g, blk, lds128, 0, A, A, C2, n, T128, C1); break;
            default: hipLaunchKernelGGL(wave_tile_kernel, dim3((nt64 + 3) / 4), blk, lds64, 0, A, B, C1, n, T64, nt64);
          }
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms / 10 < best ? ms / 10 : best;
      }
      const double ex = flops * (is_tri ? (double)tri / full : 1.0);   // EXECUTED flops
      printf("%d x %d^3  %-56s %.3f ms  %.1f TFLOP/s executed  (%.3f of 157.3)\n", nb, n, names[variant], best,
             ex / best / 1e9, ex / best / 1e9 / 157.3);
    }
    {   // the Newton square with COLD operands: six input / output sets used in rotation, so that a launch never
        // re-reads what the previous launch left in L2 / the 256 MB Infinity Cache (the stage launches of a
        // Newton step read the products of the previous launch and cycle through ten buffers per block)
      const int NS = 6;
      float* in[NS]; float* outb[NS];
      for (int q = 0; q < NS; ++q) { hipMalloc(&in[q], el * 4); hipMalloc(&outb[q], el * 4); hipMemcpy(in[q], A, el * 4, hipMemcpyDeviceToDevice); }
      for (int seg = 0; seg < 2; ++seg) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(e0);
          for (int it = 0; it < 12; ++it) {
            const dim3 g(nb * tri), blk(256);
            // chained: launch it reads what launch it - 1 wrote (the first ones read the copies of A)
            const float* src = it == 0 ? in[0] : outb[(it - 1) % NS];
            if (seg) hipLaunchKernelGGL((wg_tile_kernel<true, true, true>), g, blk, lds128, 0, src, src, outb[it % NS], n, T128, nullptr);
            else hipLaunchKernelGGL((wg_tile_kernel<true, false, true>), g, blk, lds128, 0, src, src, outb[it % NS], n, T128, nullptr);
          }
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          best = ms / 12 < best ? ms / 12 : best;
          for (int q = 0; q < NS; ++q) hipMemcpyAsync(outb[q], A, el * 4, hipMemcpyDeviceToDevice, 0);   // keep values bounded
        }
        const double ex = flops * (double)tri / full;
        printf("%d x %d^3  %-56s %.3f ms  %.1f TFLOP/s executed  (%.3f of 157.3)\n", nb, n,
               seg ? "  Newton square, CHAINED launches on rotating buffers" : "  the same without segments", best, ex / best / 1e9,
               ex / best / 1e9 / 157.3);
      }
      for (int q = 0; q < NS; ++q) { hipFree(in[q]); hipFree(outb[q]); }
    }
    {   // the Newton square again, addressed through tile entry -> block descriptor + state (as the stage kernel does)
      std::vector<MbBlock> hb(nb); std::vector<MbState> hs(nb); std::vector<MbTile> ht;
      for (int b = 0; b < nb; ++b) {
        MbBlock d{}; for (int q = 0; q < 4; ++q) d.buf[q] = A + (size_t)b * n * n;
        d.out = C2 + (size_t)b * n * n; d.n = n; d.npad = n; d.pa = 1; d.pb = 2; hb[b] = d;
        MbState z{}; z.phase = 1; z.cur = b & 1; z.it = 3; z.general = 0; hs[b] = z;
        for (int tm = 0; tm < T128; ++tm) for (int tn = tm; tn < T128; ++tn) ht.push_back({b, (short)tm, (short)tn});
      }
      MbBlock* db; MbState* dsn; MbTile* dt;
      hipMalloc(&db, hb.size() * sizeof(MbBlock)); hipMalloc(&dsn, hs.size() * sizeof(MbState)); hipMalloc(&dt, ht.size() * sizeof(MbTile));
      hipMemcpy(db, hb.data(), hb.size() * sizeof(MbBlock), hipMemcpyHostToDevice);
      hipMemcpy(dsn, hs.data(), hs.size() * sizeof(MbState), hipMemcpyHostToDevice);
      hipMemcpy(dt, ht.data(), ht.size() * sizeof(MbTile), hipMemcpyHostToDevice);
      hipFuncSetAttribute((const void*)wg_tile_indirect_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it)
          hipLaunchKernelGGL(wg_tile_indirect_kernel, dim3((unsigned)ht.size()), dim3(256), lds128, 0, db, dsn, dt, (int)ht.size());
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms / 10 < best ? ms / 10 : best;
      }
      const double ex = flops * (double)tri / full;
      printf("%d x %d^3  %-56s %.3f ms  %.1f TFLOP/s executed  (%.3f of 157.3)\n", nb, n,
             "  Newton square through tile entry -> descriptor -> state", best, ex / best / 1e9, ex / best / 1e9 / 157.3);
      hipFree(db); hipFree(dsn); hipFree(dt);
    }
    hipFree(C2);
    std::vector<float> c0(1 << 16), c1(1 << 16);
    hipMemcpy(c0.data(), C0 + el / 2, c0.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c1.data(), C1 + el / 2, c1.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0;
    for (size_t k = 0; k < c0.size(); ++k) diff += c0[k] != c1[k];
    printf("   elements that differ between the two structures (same k order expected): %zu of %zu; sample %.6f %.6f\n", diff,
           c0.size(), c0[5], c1[5]);
    hipFree(A); hipFree(B); hipFree(C0); hipFree(C1);
  }
  return 0;
}
