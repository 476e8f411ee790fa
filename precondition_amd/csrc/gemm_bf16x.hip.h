// gemm_bf16x.hip.h -- float32 products on the bf16 MFMA by operand splitting (shared by
// csrc/newton.hip and csrc/eigh_cj.hip.h).
#pragma once
#include "gemm_core.hip.h"

namespace psk {

// Three-way bf16 split of float32 operands on the bf16 MFMA (ps_options.products = bf16x6 / bf16x3;
// the update products of the eigh root path, csrc/eigh_cj.hip.h).
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): 3 x 8
// mantissa bits = the 24 of float32.  A product is accumulated in float32 from the six partial
// products whose weight is >= 2^-16 of hi*hi (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi; the
// dropped ones are below 2^-24 relative): ~2^-22 relative per product, the size of float32's own
// accumulation rounding at K ~ 1000, on v_mfma_f32_32x32x16_bf16 at 16x the rate of the exact
// float32 MFMA, i.e. 16 / 6 = 2.7x the MFMA throughput.  NOT the parity path: results are
// float32-faithful, not the float32 MFMA's bit pattern; iteration counts may differ by one where
// the 1e-6 stop threshold is within rounding.  Symmetric blocks only (both operands are read as
// rows: element (n, k) of B is taken from B[n][k]; blocks that fail the symmetry test run the
// float32 products in the same launch).  Reported separately by bench.py (`newton_bf16x6`).
typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xbf16x4 __attribute__((ext_vector_type(4)));
constexpr int XBK = 32;                  // float32 k per stage
constexpr int XLD = XBK + 8;             // bf16 per LDS row (80 bytes: conflict-free 16-byte reads)
constexpr int XPLANE = TILE * XLD;       // bf16 elements of one plane of one operand
constexpr int XNV = (TILE * XBK / 4) / NTHREADS;   // float4 per thread and operand = 4

__device__ __forceinline__ void x_load(const float* p, int ld, int mn0, int k0, int tid,
                                       f32x4 (&r)[XNV]) {
#pragma unroll
  for (int v = 0; v < XNV; ++v) {
    const int f = tid + NTHREADS * v, row = f / (XBK / 4), k4 = (f % (XBK / 4)) * 4;
    r[v] = gload4(p + (int64_t)(mn0 + row) * ld + k0 + k4);
  }
}
// registers -> the three (TERMS = 6) or two (TERMS = 3: hi, mid) bf16 planes of one operand image
template <int TERMS>
__device__ __forceinline__ void x_split_store(uint16_t* img, int tid, const f32x4 (&r)[XNV]) {
#pragma unroll
  for (int v = 0; v < XNV; ++v) {
    const int f = tid + NTHREADS * v, row = f / (XBK / 4), k4 = (f % (XBK / 4)) * 4;
    xbf16x4 hi, mid, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = r[v][e];
      const __bf16 h = (__bf16)x;
      const float r1 = x - (float)h;
      const __bf16 m = (__bf16)r1;
      hi[e] = h; mid[e] = m;
      if (TERMS == 6) lo[e] = (__bf16)(r1 - (float)m);
    }
    uint16_t* d = img + row * XLD + k4;
    *reinterpret_cast<xbf16x4*>(d) = hi;
    *reinterpret_cast<xbf16x4*>(d + XPLANE) = mid;
    if (TERMS == 6) *reinterpret_cast<xbf16x4*>(d + 2 * XPLANE) = lo;
  }
}
__device__ __forceinline__ xbf16x8 x_frag(const uint16_t* plane, int row, int k) {
  return *reinterpret_cast<const xbf16x8*>(plane + row * XLD + k);
}

// acc = A[tile rows of A] * B[tile rows of B]^T over k in [0, Kext) (Kext a multiple of 32: npad
// is a multiple of 128 and the padding is zero).  One LDS stage (61 KB: 2 workgroups per CU), two
// register sets of global loads in flight; ends with a barrier.
// TERMS = 3 (PS_PRODUCTS_BF16X3, Precision.DEFAULT): x = hi + mid only, products mid*hi + hi*mid +
// hi*hi: ~2^-16 relative per product at half the MFMA passes and two thirds of the LDS traffic.
// ACCUM: acc is added to instead of being cleared (a product whose k range comes in pieces).
template <int TERMS, bool ACCUM = false>
__device__ __forceinline__ void gemm_tile_bf16x_sym(const float* A, int lda, int am0,
                                                    const float* B, int ldb, int bn0, int Kext,
                                                    float* smem_f, f32x16 (&acc)[2][2]) {
  uint16_t* sA = reinterpret_cast<uint16_t*>(smem_f);
  uint16_t* sB = sA + 3 * XPLANE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fk = 8 * (lane >> 5);
  if (!ACCUM) zero_acc(acc);
  const int nk = Kext / XBK;
  f32x4 ra0[XNV], rb0[XNV], ra1[XNV], rb1[XNV];
  x_load(A, lda, am0, 0, tid, ra0);
  x_load(B, ldb, bn0, 0, tid, rb0);
  if (nk > 1) { x_load(A, lda, am0, XBK, tid, ra1); x_load(B, ldb, bn0, XBK, tid, rb1); }
  x_split_store<TERMS>(sA, tid, ra0);
  x_split_store<TERMS>(sB, tid, rb0);
  __syncthreads();
  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < XBK / 16; ++ks) {
      xbf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ar = wm * 64 + t * 32 + fr, br = wn * 64 + t * 32 + fr, k = ks * 16 + fk;
        ah[t] = x_frag(sA, ar, k); am[t] = x_frag(sA + XPLANE, ar, k);
        bh[t] = x_frag(sB, br, k); bm[t] = x_frag(sB + XPLANE, br, k);
        if (TERMS == 6) { al[t] = x_frag(sA + 2 * XPLANE, ar, k); bl[t] = x_frag(sB + 2 * XPLANE, br, k); }
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {   // small terms first
          f32x16 c = acc[tm][tn];
          if (TERMS == 6) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[tm], bm[tn], c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[tm], bh[tn], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bm[tn], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], c, 0, 0, 0);
          acc[tm][tn] = c;
        }
    }
  };
  for (int kt = 0; kt < nk; kt += 2) {
    // even: tile kt in LDS; set 1 holds tile kt+1; set 0 is free -> tile kt+2
    if (kt + 2 < nk) { x_load(A, lda, am0, (kt + 2) * XBK, tid, ra0); x_load(B, ldb, bn0, (kt + 2) * XBK, tid, rb0); }
    compute();
    __syncthreads();
    if (kt + 1 >= nk) break;
    x_split_store<TERMS>(sA, tid, ra1);
    x_split_store<TERMS>(sB, tid, rb1);
    __syncthreads();
    // odd: tile kt+1 in LDS; set 0 holds tile kt+2; set 1 is free -> tile kt+3
    if (kt + 3 < nk) { x_load(A, lda, am0, (kt + 3) * XBK, tid, ra1); x_load(B, ldb, bn0, (kt + 3) * XBK, tid, rb1); }
    compute();
    __syncthreads();
    if (kt + 2 < nk) {
      x_split_store<TERMS>(sA, tid, ra0);
      x_split_store<TERMS>(sB, tid, rb0);
      __syncthreads();
    }
  }
}


}  // namespace psk
