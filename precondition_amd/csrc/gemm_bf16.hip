// gemm_bf16.hip — grouped C = A * B^T on the bf16 MFMA (v_mfma_f32_32x32x16_bf16), float32
// accumulation and output, for the Frequent-Directions branch (BASELINE configs[4]: "rank-64
// updates on 4096-dim factors, bf16 MFMA"; reference: _fd_update_root DS:1123-1290, whose SVD
// at DS:1193 this build replaces by a block subspace iteration made of d x d @ d x b
// products, precondition_amd/subspace.py).
//
// Both operands are K-CONTIGUOUS bf16 arrays: A [m][k] and Bt [n][k] (the caller converts the
// float32 covariance once per update and the tall-skinny iterate once per product with
// ps_convert_f32_to_bf16, which can also transpose).  Each operand may come as ONE bf16
// array (relative precision 2^-9) or as a hi/lo PAIR (x = hi + lo, hi = bf16(x), lo =
// bf16(x - hi): 2^-17); the kernel accumulates hi*hi (+ lo*hi + hi*lo when split) into the
// same float32 accumulators.  The product is HBM-bound (a 4096 x 4096 covariance is read
// once per product, n <= 128 columns): what matters is bytes, not MFMA passes — the split
// form reads the same bytes as float32 and is ~3.4x faster than the float32 MFMA product,
// the single form halves them again.
//
// One workgroup = 256 threads = 2x2 wavefronts, 128 x 128 output tile, BK = 32 k per LDS
// stage, double buffered; LDS rows are 32 + 8 bf16 (80 bytes: 16-lane fragment reads and
// 8-lane row writes touch every bank once).  Fragments: lane l of a wavefront holds
// A[row l&31][k = 8(l>>5) + j] (j = 0..7) = one ds_read_b128 per 32x32x16 step.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "gemm_core.hip.h"

namespace psk {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// k per stage.  Round 3 tried 64 (whole 128-byte lines per operand row and stage, 64 KB of loads
// in flight per workgroup, one workgroup per CU, no split-K): 0.168 ms for the cfg5 product
// (8 x 4096^2 @ 4096 x 96) against 0.155 ms with 32 and two workgroups per CU -- the product is
// bound by the bytes a CU can take in (A rows from HBM plus the iterate's K-slices from L2,
// 3.5 MB per workgroup), not by line granularity.
constexpr int HBK = 32;
#ifndef PS_HSETS
#define PS_HSETS 2
#endif
constexpr int HSETS = PS_HSETS;   // register sets of global loads in flight per workgroup (even)
constexpr int HLD = HBK + 8;      // LDS row stride in bf16 elements (80 bytes)
constexpr int HOP = TILE * HLD;   // bf16 elements of one operand image

struct HTask {
  const uint16_t* a_hi; const uint16_t* a_lo;   // [m][k], lo may be null
  const uint16_t* b_hi; const uint16_t* b_lo;   // [n][k], lo may be null
  float* c;
  int m, n, k;
  int64_t lda, ldb, ldc;
  int ksplit, kchunk;   // split-K (deterministic two-pass), as in ps_gemm_grouped_f32
  float* partial;       // [ksplit][m][n] when ksplit > 1
  int a_tiled;          // A planes in the tile-blocked layout (ps_convert_f32_to_bf16 mode 2)
  int sym;              // c = a a^T: only tiles tm <= tn are computed, every tile also stores its mirror
  float* ptile;         // sym: [slot][nks][128][128] partial tiles of the few K-split tail tiles
};
// kchunk / nks / pslot: a tile of a symmetric product that is cut along k on its own (pslot >= 0)
struct HTile { int task; short tm, tn; int ks; int kchunk, nks, pslot; };
constexpr int SYM_TLD = TILE + 1;   // row stride of the LDS staging tile of a mirror store

__device__ inline u32x4 gload16(const uint16_t* p) { return *(const u32x4 PS_GLOBAL*)(p); }
__device__ inline u32x4 gload16_nt(const uint16_t* p) {
  return __builtin_nontemporal_load((const u32x4 PS_GLOBAL*)(p));
}

// 128 rows x HBK k of one operand array -> registers (HV x 16 bytes per thread)
constexpr int HV = TILE * HBK / 8 / 256;   // 16-byte chunks per thread
constexpr int HCPR = HBK / 8;              // chunks per row
__device__ inline void hload(const uint16_t* base, int64_t ld, int row0, int rows, int k0,
                             int tid, u32x4 (&r)[HV]) {
#pragma unroll
  for (int v = 0; v < HV; ++v) {
    const int f = tid + 256 * v;
    const int row = f / HCPR, kc = (f % HCPR) * 8;
    // unconditional load of a clamped row, masked afterwards: a predicated load is compiled as a
    // branch around it, and with loads on conditional paths the compiler's vmcnt waits fall back
    // to vmcnt(0) -- no second register set would ever be in flight
    const int rr = min(row0 + row, rows - 1);
    const unsigned keep = row0 + row < rows ? 0xffffffffu : 0u;
    u32x4 t = gload16(base + (int64_t)rr * ld + k0 + kc);
    t.x &= keep; t.y &= keep; t.z &= keep; t.w &= keep;
    r[v] = t;
  }
}
__device__ inline void hstore(uint16_t* s, int tid, const u32x4 (&r)[HV]) {
#pragma unroll
  for (int v = 0; v < HV; ++v) {
    const int f = tid + 256 * v;
    const int row = f / HCPR, kc = (f % HCPR) * 8;
    *reinterpret_cast<u32x4*>(s + row * HLD + kc) = r[v];
  }
}
__device__ inline bf16x8 hfrag(const uint16_t* s, int row, int k) {
  return *reinterpret_cast<const bf16x8*>(s + row * HLD + k);
}

// SA / SB: 1 = single bf16 array, 2 = hi/lo pair.
template <int SA, int SB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_grouped_kernel(const HTask* tasks,
                                                                   const HTile* tiles,
                                                                   int ntiles) {
  extern __shared__ __align__(16) uint16_t hs[];   // 2 stages x (SA + SB) images
  constexpr int STG = (SA + SB) * HOP;
  const HTile te = tiles[xcd_remap(blockIdx.x, ntiles)];
  const HTask tk = tasks[te.task];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int row0 = te.tm * TILE, col0 = te.tn * TILE;
  const int kchunk = te.pslot >= 0 ? te.kchunk : tk.kchunk;
  const int kbeg = te.ks * kchunk;                    // k % 32 == 0 and kchunk % 32 == 0
  const int nk = (min(kchunk, tk.k - kbeg) + HBK - 1) / HBK;      // (checked on the host)
  f32x16 acc[2][2];
  zero_acc(acc);
  u32x4 ra[HSETS][SA][HV], rb[HSETS][SB][HV];
  // Tile-blocked A (the 4096^2 covariance of the FD branch): tile (tm, kt) is 128 x 32 bf16 =
  // 8 KB CONTIGUOUS, so a workgroup streams whole DRAM pages; row-major, every K-tile touches
  // 128 half cache lines that lie 8 KB apart (65536 concurrent row streams over the chip: the
  // product ran at 0.44 of the HBM peak whatever the prefetch depth).  Same loader: a tile is a
  // 128 x 32 matrix with leading dimension 32 (rows beyond m are zero padding in the buffer).
  const bool at = tk.a_tiled != 0;
  const int64_t a_ld = at ? HBK : tk.lda;
  const int a_row0 = at ? 0 : row0, a_rows = at ? TILE : tk.m;
  const int64_t a_tile0 = at ? (int64_t)te.tm * (tk.k / HBK) * (TILE * HBK) : 0;
  const uint16_t* a_hi = tk.a_hi + a_tile0;
  const uint16_t* a_lo = SA == 2 ? tk.a_lo + a_tile0 : nullptr;
  // element offset of K-tile k0 / HBK inside the row panel
  auto a_off = [&](int k0) { return at ? (int64_t)(k0 / HBK) * (TILE * HBK) : (int64_t)0; };
  auto a_k = [&](int k0) { return at ? 0 : k0; };
  // HSETS register sets of global loads in flight (tile t lives in set t % HSETS from its request,
  // issued when that set was last written to LDS, until it is written to the LDS stage at the
  // bottom of K-tile t - 1).  Requests past the end re-read the last K-tile: unconditional loads
  // keep the compiler's vmcnt exact.  Measured on the cfg5 product (8 x 4096^2 @ 4096 x 96, hi/lo):
  // 1, 2 or 4 sets make no difference (0.150-0.156 ms) -- the product is not bound by the
  // latency a workgroup can cover; the tile-blocked A layout is what moved it (0.167 -> 0.150).
  auto load_set = [&](int sidx, int t) {
    const int k0 = kbeg + min(t, nk - 1) * HBK;
    hload(a_hi + a_off(k0), a_ld, a_row0, a_rows, a_k(k0), tid, ra[sidx][0]);
    if (SA == 2) hload(a_lo + a_off(k0), a_ld, a_row0, a_rows, a_k(k0), tid, ra[sidx][SA - 1]);
    hload(tk.b_hi, tk.ldb, col0, tk.n, k0, tid, rb[sidx][0]);
    if (SB == 2) hload(tk.b_lo, tk.ldb, col0, tk.n, k0, tid, rb[sidx][SB - 1]);
  };
  auto store_set = [&](int sidx, uint16_t* dst) {
    hstore(dst, tid, ra[sidx][0]);
    if (SA == 2) hstore(dst + HOP, tid, ra[sidx][SA - 1]);
    hstore(dst + SA * HOP, tid, rb[sidx][0]);
    if (SB == 2) hstore(dst + (SA + 1) * HOP, tid, rb[sidx][SB - 1]);
  };
#pragma unroll
  for (int u = 0; u < HSETS; ++u) load_set(u, u);
  store_set(0, hs);
  __syncthreads();
  const int fr = lane & 31, fk = 8 * (lane >> 5);
  for (int kt = 0; kt < nk; kt += HSETS) {
#pragma unroll
    for (int u = 0; u < HSETS; ++u) {
      const int t = kt + u;
      if (t >= nk) break;
      uint16_t* cur = hs + (u & 1) * STG;          // HSETS is even: t & 1 == u & 1
      uint16_t* nxt = hs + ((u + 1) & 1) * STG;
      load_set(u, t + HSETS);                      // set u was written to LDS one K-tile ago
#pragma unroll
      for (int ks = 0; ks < HBK / 16; ++ks) {
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          ah[q] = hfrag(cur, wm * 64 + q * 32 + fr, ks * 16 + fk);
          if (SA == 2) al[q] = hfrag(cur + HOP, wm * 64 + q * 32 + fr, ks * 16 + fk);
          bh[q] = hfrag(cur + SA * HOP, wn * 64 + q * 32 + fr, ks * 16 + fk);
          if (SB == 2) bl[q] = hfrag(cur + (SA + 1) * HOP, wn * 64 + q * 32 + fr, ks * 16 + fk);
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            // small terms first: lo*hi + hi*lo, then hi*hi (lo*lo ~ 2^-18 relative is dropped)
            if (SA == 2)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            if (SB == 2)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
          }
      }
      if (t + 1 < nk) store_set((u + 1) % HSETS, nxt);
      __syncthreads();
    }
  }
  if (te.pslot >= 0) {
    // K-split tail tile of a symmetric product: the whole 128 x 128 partial tile, compact
    float* pt = tk.ptile + ((int64_t)te.pslot * te.nks + te.ks) * (TILE * TILE);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          gstore1(pt + acc_row(wm, i, r, lane) * TILE + acc_col(wn, j, lane), acc[i][j][r]);
    return;
  }
  float* out = tk.ksplit > 1 ? tk.partial + (int64_t)te.ks * tk.m * tk.n : tk.c;
  const int64_t ldo = tk.ksplit > 1 ? tk.n : tk.ldc;
  const bool diag = tk.sym && te.tm == te.tn;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + acc_row(wm, i, r, lane);
        const int col = col0 + acc_col(wn, j, lane);
        if (row < tk.m && col < tk.n && (!diag || row <= col)) {
          gstore1(out + (int64_t)row * ldo + col, acc[i][j][r]);
          // a diagonal tile keeps its upper triangle and mirrors it: hi*lo and lo*hi enter the
          // accumulation chain in a fixed order, so (i, j) and (j, i) differ in the last bits
          if (diag && row < col) gstore1(out + (int64_t)col * ldo + row, acc[i][j][r]);
        }
      }
  if (!tk.sym || diag) return;
  // mirror tile (tn, tm): through LDS, so that the global stores run along rows
  float* st = reinterpret_cast<float*>(hs);
  __syncthreads();                               // the K loop's last fragment reads are done
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        st[acc_row(wm, i, r, lane) * SYM_TLD + acc_col(wn, j, lane)] = acc[i][j][r];
  __syncthreads();
  for (int e = tid; e < TILE * TILE; e += 256) {
    const int c = e >> 7, r = e & (TILE - 1);     // mirror row = col0 + c, mirror column = row0 + r
    if (col0 + c < tk.n && row0 + r < tk.m)
      gstore1(tk.c + (int64_t)(col0 + c) * tk.ldc + row0 + r, st[r * SYM_TLD + c]);
  }
}

// K-split tiles of a symmetric product: sum of the nks partial tiles in split order, stored as the
// tile and as its mirror.  SYM_RED workgroups per tile (16 tile rows each: with one workgroup per
// tile the 16 tail tiles of a 4096^2 Gram matrix took 160 us, as long as half the product).
struct HSymSlot { int task; short tm, tn; int nks, local; };   // local: index among its task's slots
constexpr int SYM_RED = 8, SYM_RR = TILE / SYM_RED;
__global__ __launch_bounds__(256) void gemm_bf16_symtile_reduce_kernel(const HTask* tasks,
                                                                      const HSymSlot* slots) {
  __shared__ float st[SYM_RR * (TILE + 1)];
  const HSymSlot sl = slots[blockIdx.x / SYM_RED];
  const int part = blockIdx.x % SYM_RED;
  const HTask tk = tasks[sl.task];
  const float* pt = tk.ptile + (int64_t)sl.local * sl.nks * (TILE * TILE);
  const int row0 = sl.tm * TILE, col0 = sl.tn * TILE, tid = threadIdx.x;
  const bool diag = sl.tm == sl.tn;
  for (int e = tid; e < SYM_RR * TILE; e += 256) {
    const int lr = e >> 7, r = part * SYM_RR + lr, c = e & (TILE - 1);
    const float* q0 = pt + r * TILE + c;
    float v = 0.f;
    int q = 0;
    for (; q + 4 <= sl.nks; q += 4) {     // four partials in flight, summed in split order
      const float p0 = gload1(q0 + (int64_t)(q + 0) * TILE * TILE), p1 = gload1(q0 + (int64_t)(q + 1) * TILE * TILE),
                  p2 = gload1(q0 + (int64_t)(q + 2) * TILE * TILE), p3 = gload1(q0 + (int64_t)(q + 3) * TILE * TILE);
      v = (((v + p0) + p1) + p2) + p3;
    }
    for (; q < sl.nks; ++q) v += gload1(q0 + (int64_t)q * TILE * TILE);
    // a diagonal tile keeps its upper triangle and mirrors it (see gemm_bf16_grouped_kernel)
    if (row0 + r < tk.m && col0 + c < tk.n && (!diag || r <= c))
      gstore1(tk.c + (int64_t)(row0 + r) * tk.ldc + col0 + c, v);
    st[lr * (TILE + 1) + c] = v;
  }
  __syncthreads();
  for (int e = tid; e < SYM_RR * TILE; e += 256) {
    const int c = e / SYM_RR, lr = e % SYM_RR, r = part * SYM_RR + lr;   // mirror row col0 + c, column row0 + r
    if (col0 + c < tk.n && row0 + r < tk.m && (!diag || r < c))
      gstore1(tk.c + (int64_t)(col0 + c) * tk.ldc + row0 + r, st[lr * (TILE + 1) + c]);
  }
}

__global__ __launch_bounds__(256) void gemm_bf16_splitk_reduce_kernel(const HTask* tasks) {
  const HTask tk = tasks[blockIdx.x];
  if (tk.ksplit <= 1) return;
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

// ---- float32 -> bf16 (round to nearest even), optionally hi/lo split and transposed -------
// dst_hi[r][c] = bf16(src[r][c]); dst_lo[r][c] = bf16(src[r][c] - float(dst_hi[r][c])).
// transpose: dst[c][r] instead (64 x 64 tiles through LDS, both sides coalesced).
__global__ __launch_bounds__(256) void cvt_bf16_kernel(const float* src, uint16_t* hi,
                                                       uint16_t* lo, int rows, int cols,
                                                       int64_t lds, int64_t ldd, int transpose,
                                                       int tiles_c, uint16_t* lo2 = nullptr) {
  __shared__ float t[64][65];
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
  const int r0 = tr * 64, c0 = tc * 64, tid = threadIdx.x;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    t[r][c] = (r0 + r < rows && c0 + c < cols) ? gload1(src + (int64_t)(r0 + r) * lds + c0 + c) : 0.f;
  }
  __syncthreads();
  if (transpose == 3) {
    // fragment-major left operand of fd_cy_step_kernel: the 64 x 64 tile is ONE contiguous run of
    // 4096 elements, [k block of 16][row half of 32][lane = 32 (k / 8 % 2) + row % 32][8 k]
    const int64_t base = ((int64_t)tr * (cols / 16) + c0 / 16) * 1024;
    for (int e = tid; e < 64 * 64; e += 256) {
      const int kk = e >> 10, sub = (e >> 9) & 1, ln = (e >> 3) & 63, ki = e & 7;
      const float x = t[sub * 32 + (ln & 31)][kk * 16 + (ln >> 5) * 8 + ki];
      const __bf16 h = (__bf16)x;
      hi[base + e] = __builtin_bit_cast(uint16_t, h);
      if (lo != nullptr) {
        const __bf16 l = (__bf16)(x - (float)h);
        lo[base + e] = __builtin_bit_cast(uint16_t, l);
        // third plane of the six-product form (ps_fd_cx6_f32): x = hi + lo + lo2 to float32 accuracy
        if (lo2 != nullptr) lo2[base + e] = __builtin_bit_cast(uint16_t, (__bf16)((x - (float)h) - (float)l));
      }
    }
    return;
  }
  for (int e = tid; e < 64 * 64; e += 256) {
    const int a = e >> 6, b = e & 63;          // output tile coordinates (row a, col b)
    const int r = transpose == 1 ? b : a, c = transpose == 1 ? a : b;
    if (r0 + r >= rows || c0 + c >= cols) continue;
    const float x = t[r][c];
    const __bf16 h = (__bf16)x;                 // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    int64_t o = transpose == 1 ? (int64_t)(c0 + c) * ldd + r0 + r : (int64_t)(r0 + r) * ldd + c0 + c;
    if (transpose == 2) {   // tile-blocked: [row tile of 128][k tile of 32][128][32]
      const int gr = r0 + r, gc = c0 + c;
      o = ((int64_t)(gr / TILE) * (cols / HBK) + gc / HBK) * (TILE * HBK) + (gr % TILE) * HBK + gc % HBK;
    }
    hi[o] = __builtin_bit_cast(uint16_t, h);
    if (lo != nullptr) {
      const __bf16 l = (__bf16)(x - (float)h);
      lo[o] = __builtin_bit_cast(uint16_t, l);
    }
  }
}


// ---- one whole step of the Chebyshev filter of the FD branch in ONE launch ---------------------
//   z = C y (bf16 hi/lo operands, three products, float32 accumulation)
//   y' = (z - ctr y) 2 sigma' / e - sigma sigma' y_prev          (ps_fd_filter_step_f32, step >= 2)
//   y' also as the bf16 hi/lo operand of the next step
// The product is a stream over C (4 bytes per element, read once) against a block of b <= 96
// columns: nothing of C is reused, so it does not go through LDS.  Both operands are stored
// FRAGMENT-MAJOR -- the 64 x 16 bytes a wavefront feeds to one v_mfma_f32_32x32x16_bf16 are one
// contiguous kilobyte (lane l: row or column l % 32, k = 8 (l / 32) .. + 7):
//   C   [n / 64][n / 16][2][64][8]   (ps_convert_f32_to_bf16, transpose = 3)
//   Y^T [n / 16][b / 32][64][8]      per factor (written by the previous step)
// One workgroup = 64 rows of one factor; its four wavefronts take the 16-wide k blocks round robin
// (kk = w, w + 4, ...: together they walk C's rows as one sequential stream of 8 KB per round),
// keep 2 x (b / 32) accumulator tiles each, and add them through LDS in a fixed order at the end.  No barrier and no LDS traffic inside the K loop; two register sets of loads in flight.
// The iterate planes are read from L2 (all 64 workgroups of a factor run on one XCD, xcd_remap).
struct FdCyArgs {
  const uint16_t* a_hi[16]; const uint16_t* a_lo[16];
  const uint16_t* bt_hi; const uint16_t* bt_lo;
  const float* y; const float* y_prev; float* y_next;
  uint16_t* nt_hi; uint16_t* nt_lo;
  const float* params;
  int step, n, nwg;
};

// SUB = 32-row halves per workgroup: 2 (64 rows) when the factors fill the chip, 1 for one or two factors
// (twice the workgroups: a single 4096-dim factor is only 64 blocks of 64 rows).
template <int CB, int SUB = 2>
__global__ __launch_bounds__(256, 2) void fd_cy_step_kernel(const FdCyArgs a) {
  constexpr int B = CB * 32, ZLD = B + 1, ROWS = 32 * SUB, NQ = ROWS * B / 4 / 256;
  __shared__ float zt[2][ROWS * ZLD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = a.n / ROWS;
  const int t = xcd_remap(blockIdx.x, a.nwg);
  const int j = t / per, rb = t % per, r0 = rb * ROWS;
  const float ctr = a.params[4 * j + 0], e = a.params[4 * j + 1], sigma1 = a.params[4 * j + 2];
  const int deg = (int)a.params[4 * j + 3];
  const int64_t fb = (int64_t)j * a.n * B;            // factor base (elements) in y* and the planes
  const int64_t blk = fb + (int64_t)r0 * B;           // this workgroup's ROWS x B block: contiguous
  if (a.step > deg) {                                 // the factor's filter is finished: y' = y
    for (int q = tid; q < ROWS * B / 4; q += 256)
      *(f32x4 PS_GLOBAL*)(a.y_next + blk + 4 * q) = gload4(a.y + blk + 4 * q);
    return;
  }
  float sigma = sigma1;
  for (int m = 2; m < a.step; ++m) sigma = 1.f / (2.f / sigma1 - sigma);
  const float sn = 1.f / (2.f / sigma1 - sigma);
  const float c1 = 2.f * sn / e, c2 = sigma * sn;

  const int nkk = a.n >> 4, cnt = nkk >> 2;           // k blocks of 16; per wavefront (n % 128 == 0)
  // C planes: [n / 64][n / 16][2 halves][64 lanes][8]; a 32-row workgroup reads one half of every kilobyte pair
  const int64_t a0 = (((int64_t)(r0 >> 6) * nkk * 2 + ((r0 >> 5) & 1)) * 64 + lane) * 8;
  const uint16_t* pah = a.a_hi[j] + a0;
  const uint16_t* pal = a.a_lo[j] + a0;
  const uint16_t* pbh = a.bt_hi + fb + lane * 8;
  const uint16_t* pbl = a.bt_lo + fb + lane * 8;
  // this block of y and y_prev (the recurrence's other inputs): requested now, used after the K loop
  f32x4 yy[NQ], yp[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    yy[q] = gload4(a.y + blk + 4 * (tid + 256 * q));
    yp[q] = gload4(a.y_prev + blk + 4 * (tid + 256 * q));
  }
  __builtin_amdgcn_sched_barrier(0);
  u32x4 ra[2][2 * SUB], rb_[2][2 * CB];
  auto load_set = [&](int s, int i) {
    const int kk = w + 4 * i;
    const uint16_t* qh = pah + (int64_t)kk * 1024;
    const uint16_t* ql = pal + (int64_t)kk * 1024;
    // C is read once: non-temporal, so that it does not push the iterate planes (re-read by every
    // workgroup of the factor) out of the XCD's L2 (137 -> 126 us per step at 8 x 4096^2)
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      ra[s][sub] = gload16_nt(qh + 512 * sub);
      ra[s][SUB + sub] = gload16_nt(ql + 512 * sub);
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      rb_[s][cb] = gload16(pbh + (int64_t)(kk * CB + cb) * 512);
      rb_[s][CB + cb] = gload16(pbl + (int64_t)(kk * CB + cb) * 512);
    }
  };
  f32x16 acc[SUB][CB];
#pragma unroll
  for (int sub = 0; sub < SUB; ++sub)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[sub][cb][r] = 0.f;
  auto compute = [&](int s) {
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, ra[s][sub]);
      const bf16x8 al = __builtin_bit_cast(bf16x8, ra[s][SUB + sub]);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, rb_[s][cb]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, rb_[s][CB + cb]);
        // small terms first: lo*hi + hi*lo, then hi*hi (lo*lo ~ 2^-18 relative is dropped)
        acc[sub][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[sub][cb], 0, 0, 0);
        acc[sub][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[sub][cb], 0, 0, 0);
        acc[sub][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[sub][cb], 0, 0, 0);
      }
    }
  };
  load_set(0, 0);
  __builtin_amdgcn_sched_barrier(0);   // set 0 must be requested first: the loop's waits count loads in order
  load_set(1, 1);
  __builtin_amdgcn_sched_barrier(0);
  // sched_barrier: without it the scheduler sinks every load to just before its use (fewer live
  // registers, but a vmcnt(0) wait after each handful of loads: nothing stays in flight)
  for (int i = 0; i < cnt - 2; i += 2) {              // cnt is even and >= 2
    compute(0);
    __builtin_amdgcn_sched_barrier(0);
    load_set(0, i + 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    __builtin_amdgcn_sched_barrier(0);
    load_set(1, i + 3);
    __builtin_amdgcn_sched_barrier(0);
  }
  compute(0);
  compute(1);
  // z tile = (w0 + w2) + (w1 + w3) of the four wavefronts' accumulators (fixed order): wavefronts
  // 0 / 1 store into the two LDS tiles, 2 / 3 add theirs
  for (int p = 0; p < 2; ++p) {
    if ((w >> 1) == p) {
      float* zw = zt[w & 1];
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float* q = zw + row * ZLD + cb * 32 + (lane & 31);
            *q = p == 0 ? acc[sub][cb][r] : *q + acc[sub][cb][r];
          }
    }
    __syncthreads();
  }
  // recurrence on the ROWS x B block (contiguous in y, y_prev, y_next)
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int q = tid + 256 * qi;
    const int r = (q * 4) / B, c = (q * 4) % B;
    float* zr = zt[0] + r * ZLD + c;
    const float* zs = zt[1] + r * ZLD + c;
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // the arithmetic of fd_filter_step_kernel, element for element
      float x = ((zr[i] + zs[i]) - ctr * yy[qi][i]) * c1;
      x -= c2 * yp[qi][i];
      v[i] = x;
      zr[i] = x;
    }
    *(f32x4 PS_GLOBAL*)(a.y_next + blk + 4 * q) = v;
  }
  if (a.nt_hi == nullptr) return;
  __syncthreads();
  // the new iterate as the next step's fragment-major bf16 operand: this block's 64 rows are k
  // blocks r0 / 16 .. + 3 of the factor's planes = one contiguous run of 4 * CB kilobytes per plane
  for (int it = tid; it < (ROWS / 16) * CB * 64; it += 256) {
    const int ln = it & 63, cb = (it >> 6) % CB, kk = (it >> 6) / CB;
    const float* col = zt[0] + (kk * 16 + (ln >> 5) * 8) * ZLD + cb * 32 + (ln & 31);
    u32x4 h4, l4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float x0 = col[(2 * i) * ZLD], x1 = col[(2 * i + 1) * ZLD];
      const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
      const __bf16 l0 = (__bf16)(x0 - (float)h0), l1 = (__bf16)(x1 - (float)h1);
      h4[i] = (unsigned)__builtin_bit_cast(uint16_t, h0) | ((unsigned)__builtin_bit_cast(uint16_t, h1) << 16);
      l4[i] = (unsigned)__builtin_bit_cast(uint16_t, l0) | ((unsigned)__builtin_bit_cast(uint16_t, l1) << 16);
    }
    const int64_t o = fb + ((int64_t)((r0 >> 4) + kk) * CB + cb) * 512 + ln * 8;
    *(u32x4 PS_GLOBAL*)(a.nt_hi + o) = h4;
    *(u32x4 PS_GLOBAL*)(a.nt_lo + o) = l4;
  }
}


// ---- Z = C X to float32 accuracy on the bf16 MFMA (the Rayleigh-Ritz product of the FD branch) -----
// Both operands as THREE bf16 planes (x = p0 + p1 + p2, 8 + 8 + 8 mantissa bits) and the six products
// whose weight is above 2^-24: p2 q0 + p1 q1 + p0 q2 + p1 q0 + p0 q1 + p0 q0 (smallest first), float32
// accumulation -- the arithmetic of the Newton path's bf16x6 mode (gemm_bf16x.hip.h).  Same structure
// as fd_cy_step_kernel (fragment-major planes, no LDS in the K loop): the float32 MFMA product it
// replaces is bound by the float32 matrix rate (0.39 ms for 8 x 4096^2 @ 4096 x 96), this one by the
// 6 bytes per element of C it streams.
struct FdCx6Args {
  const uint16_t* a[3][16];
  const uint16_t* bt[3];
  float* z;
  int n, nwg;
};

// x [B][n][b] float32 -> three fragment-major planes (the Y^T layout of fd_cy_step_kernel)
template <int CB>
__global__ __launch_bounds__(256) void fd_split3_frag_kernel(const float* x, uint16_t* p0, uint16_t* p1,
                                                             uint16_t* p2, int n) {
  constexpr int B = CB * 32, ZLD = B + 1;
  __shared__ float t[64 * ZLD];
  const int per = n >> 6, j = blockIdx.x / per, rb = blockIdx.x % per, r0 = rb * 64, tid = threadIdx.x;
  const int64_t fb = (int64_t)j * n * B, blk = fb + (int64_t)r0 * B;
  for (int q = tid; q < 64 * B / 4; q += 256) {
    const f32x4 v = gload4(x + blk + 4 * q);
    float* d = t + ((q * 4) / B) * ZLD + (q * 4) % B;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
  __syncthreads();
  for (int it = tid; it < 4 * CB * 64; it += 256) {
    const int ln = it & 63, cb = (it >> 6) % CB, kk = (it >> 6) / CB;
    const float* col = t + (kk * 16 + (ln >> 5) * 8) * ZLD + cb * 32 + (ln & 31);
    u32x4 w0, w1, w2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned u0 = 0, u1 = 0, u2 = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float v = col[(2 * i + h) * ZLD];
        const __bf16 a = (__bf16)v;
        const __bf16 bb = (__bf16)(v - (float)a);
        const __bf16 c = (__bf16)((v - (float)a) - (float)bb);
        u0 |= (unsigned)__builtin_bit_cast(uint16_t, a) << (16 * h);
        u1 |= (unsigned)__builtin_bit_cast(uint16_t, bb) << (16 * h);
        u2 |= (unsigned)__builtin_bit_cast(uint16_t, c) << (16 * h);
      }
      w0[i] = u0; w1[i] = u1; w2[i] = u2;
    }
    const int64_t o = fb + ((int64_t)((r0 >> 4) + kk) * CB + cb) * 512 + ln * 8;
    *(u32x4 PS_GLOBAL*)(p0 + o) = w0;
    *(u32x4 PS_GLOBAL*)(p1 + o) = w1;
    *(u32x4 PS_GLOBAL*)(p2 + o) = w2;
  }
}

template <int CB, int SUB = 2>
__global__ __launch_bounds__(256, 2) void fd_cx6_kernel(const FdCx6Args a) {
  constexpr int B = CB * 32, ZLD = B + 1, ROWS = 32 * SUB, NQ = ROWS * B / 4 / 256;
  __shared__ float zt[2][ROWS * ZLD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = a.n / ROWS;
  const int t = xcd_remap(blockIdx.x, a.nwg);
  const int j = t / per, rb = t % per, r0 = rb * ROWS;
  const int64_t fb = (int64_t)j * a.n * B, blk = fb + (int64_t)r0 * B;
  const int nkk = a.n >> 4, cnt = nkk >> 2;
  const uint16_t* pa[3];
  const uint16_t* pb[3];
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    pa[pl] = a.a[pl][j] + (((int64_t)(r0 >> 6) * nkk * 2 + ((r0 >> 5) & 1)) * 64 + lane) * 8;
    pb[pl] = a.bt[pl] + fb + lane * 8;
  }
  u32x4 ra[2][3][SUB], rb_[2][3][CB];
  auto load_set = [&](int s, int i) {
    const int kk = w + 4 * i;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) ra[s][pl][sub] = gload16_nt(pa[pl] + (int64_t)kk * 1024 + 512 * sub);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) rb_[s][pl][cb] = gload16(pb[pl] + (int64_t)(kk * CB + cb) * 512);
  };
  f32x16 acc[SUB][CB];
#pragma unroll
  for (int sub = 0; sub < SUB; ++sub)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[sub][cb][r] = 0.f;
  auto compute = [&](int s) {
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      const bf16x8 a0 = __builtin_bit_cast(bf16x8, ra[s][0][sub]);
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, ra[s][1][sub]);
      const bf16x8 a2 = __builtin_bit_cast(bf16x8, ra[s][2][sub]);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, rb_[s][0][cb]);
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, rb_[s][1][cb]);
        const bf16x8 b2 = __builtin_bit_cast(bf16x8, rb_[s][2][cb]);
        acc[sub][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[sub][cb], 0, 0, 0);
        acc[sub][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[sub][cb], 0, 0, 0);
        acc[sub][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[sub][cb], 0, 0, 0);
        acc[sub][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[sub][cb], 0, 0, 0);
        acc[sub][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[sub][cb], 0, 0, 0);
        acc[sub][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[sub][cb], 0, 0, 0);
      }
    }
  };
  load_set(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  load_set(1, 1);
  __builtin_amdgcn_sched_barrier(0);
  for (int i = 0; i < cnt - 2; i += 2) {
    compute(0);
    __builtin_amdgcn_sched_barrier(0);
    load_set(0, i + 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    __builtin_amdgcn_sched_barrier(0);
    load_set(1, i + 3);
    __builtin_amdgcn_sched_barrier(0);
  }
  compute(0);
  compute(1);
  for (int p = 0; p < 2; ++p) {
    if ((w >> 1) == p) {
      float* zw = zt[w & 1];
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float* q = zw + row * ZLD + cb * 32 + (lane & 31);
            *q = p == 0 ? acc[sub][cb][r] : *q + acc[sub][cb][r];
          }
    }
    __syncthreads();
  }
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int q = tid + 256 * qi;
    const int r = (q * 4) / B, c = (q * 4) % B;
    const float* zr = zt[0] + r * ZLD + c;
    const float* zs = zt[1] + r * ZLD + c;
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = zr[i] + zs[i];
    *(f32x4 PS_GLOBAL*)(a.z + blk + 4 * q) = v;
  }
}

}  // namespace psk



using namespace psk;

extern "C" int ps_convert_f32_to_bf16(void* stream, const float* src, void* dst_hi,
                                      void* dst_lo, int64_t rows, int64_t cols, int64_t lds,
                                      int64_t ldd, int transpose) {
  PS_DEVICE_CHECK();
  if (!src || !dst_hi || rows < 1 || cols < 1 || lds < cols || transpose < 0 || transpose > 3 ||
      (transpose < 2 && ldd < (transpose ? rows : cols)))
    return PS_EINVAL;
  if (transpose == 3 && (rows % 64 != 0 || cols % 64 != 0)) return PS_EUNSUPPORTED;
  if (transpose == 2) {
    // tile-blocked operand of ps_gemm_bf16_grouped (a_tiled): the destination holds
    // ceil(rows / 128) * 128 * cols elements; rows past `rows` must read as zero
    if (cols % HBK != 0) return PS_EUNSUPPORTED;
    const size_t bytes = (size_t)((rows + TILE - 1) / TILE) * TILE * (size_t)cols * sizeof(uint16_t);
    if (rows % TILE != 0) {
      PS_HIP(hipMemsetAsync(dst_hi, 0, bytes, (hipStream_t)stream));
      if (dst_lo) PS_HIP(hipMemsetAsync(dst_lo, 0, bytes, (hipStream_t)stream));
    }
  }
  const int64_t tr = (rows + 63) / 64, tc = (cols + 63) / 64;
  if (tr * tc > 0x7fffffff) return PS_EUNSUPPORTED;
  hipLaunchKernelGGL(cvt_bf16_kernel, dim3((unsigned)(tr * tc)), dim3(256), 0,
                     (hipStream_t)stream, src, (uint16_t*)dst_hi, (uint16_t*)dst_lo, (int)rows,
                     (int)cols, lds, ldd, transpose, (int)tc);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

static int hsplit_for(size_t total_tiles, int k) {
  if (total_tiles >= 512 || k < 1024) return 1;
  int s = (int)((512 + total_tiles - 1) / total_tiles);   // 2 resident workgroups per CU
  s = std::min(s, k / 512);
  return std::max(1, std::min(s, 16));
}

static size_t htiles(const ps_gemm_bf16_desc* d, int count) {
  size_t tiles = 0;
  for (int i = 0; i < count; ++i)
    tiles += (size_t)((d[i].m + TILE - 1) / TILE) * ((d[i].n + TILE - 1) / TILE);
  return tiles;
}

// Symmetric products (c = a a^T): the tiles with tm <= tn, diagonal ones first.  With T (T + 1) / 2
// equal tiles on a chip that holds SYM_SLOTS of them at a time the last partial round would take as
// long as a full one (528 tiles of a 4096^2 Gram matrix: 512 + 16), so the tiles of that round
// -- when they are few -- are cut along k into pieces (compact partial tiles, summed and mirrored
// by gemm_bf16_symtile_reduce_kernel): the round then takes 1 / 32 of a tile.  A product with fewer
// tiles than slots is cut in the same way as a whole (36 tiles of a 1024^2 Gram matrix: 15 pieces each).
constexpr int SYM_SLOTS = 512, SYM_TAIL_MAX = 64, SYM_NKS = 32;
struct SymPlan { int64_t tiles; int tail, kc, nks; };   // the last `tail` tiles in nks pieces of kc
static SymPlan sym_plan(int m, int k) {
  const int T = (m + TILE - 1) / TILE;
  SymPlan p{(int64_t)T * (T + 1) / 2, 0, k, 1};
  int want = 1;
  if (p.tiles < SYM_SLOTS) {
    p.tail = (int)p.tiles;
    want = std::min<int64_t>(SYM_NKS, (SYM_SLOTS + p.tiles - 1) / p.tiles);
  } else if (p.tiles % SYM_SLOTS != 0 && p.tiles % SYM_SLOTS <= SYM_TAIL_MAX) {
    p.tail = (int)(p.tiles % SYM_SLOTS);
    want = SYM_NKS;
  }
  // k per piece: a multiple of HBK, at least 4 K-tiles; nks = the number of NON-EMPTY pieces
  p.kc = std::max(4 * HBK, (int)psh::round_up((k + want - 1) / want, HBK));
  p.nks = (k + p.kc - 1) / p.kc;
  if (p.nks <= 1) { p.tail = 0; p.nks = 1; p.kc = k; }
  return p;
}

static size_t hbytes(const ps_gemm_bf16_desc* d, int count) {
  const size_t tiles = htiles(d, count);
  size_t split_tiles = 0, partial = 0;
  for (int i = 0; i < count; ++i) {
    const int s = d[i].symmetric ? 1 : hsplit_for(tiles, d[i].k);
    const size_t t = (size_t)((d[i].m + TILE - 1) / TILE) * ((d[i].n + TILE - 1) / TILE);
    split_tiles += (size_t)s * t;
    if (s > 1) partial += psh::align_up((size_t)s * d[i].m * d[i].n * sizeof(float), 256) + 256;
    if (d[i].symmetric) {
      const SymPlan sp = sym_plan(d[i].m, d[i].k);
      split_tiles += (size_t)sp.tail * sp.nks;
      partial += psh::align_up((size_t)sp.tail * sp.nks * TILE * TILE * sizeof(float), 256) + 256 +
                 4 * (psh::align_up(sizeof(HSymSlot) * sp.tail, 256) + 256);
    }
  }
  return 4 * (psh::align_up(sizeof(HTask) * count, 256) + 256) +
         4 * (psh::align_up(sizeof(HTile) * split_tiles, 256) + 256) + partial + 1024;
}

extern "C" size_t ps_gemm_bf16_grouped_workspace_bytes(const ps_gemm_bf16_desc* desc, int count) {
  if (!desc || count <= 0) return 0;
  return hbytes(desc, count);
}

// A grouped product as a reusable plan: the task / tile tables are uploaded once (hplan_build) and
// may be launched any number of times while the operand POINTERS stay the same (hplan_launch) --
// the Chebyshev filter of the FD branch multiplies the same covariances by an iterate that is
// rewritten in place ~12 times per round (ps_fd_filter_round_f32).
struct HPlan {
  struct Group {
    HTask* dt = nullptr; HTile* dl = nullptr; int nt = 0, ntasks = 0; bool any_split = false;
    HSymSlot* ds = nullptr; int nslots = 0; bool any_sym = false;
  };
  Group g[4];
};

static int hplan_build(hipStream_t st, const ps_gemm_bf16_desc* desc, int count, void* workspace,
                       size_t workspace_bytes, HPlan& pl) {
  if (!desc || count <= 0 || !workspace) return PS_EINVAL;
  if (workspace_bytes < hbytes(desc, count)) return PS_EWORKSPACE;
  // groups by (split of A, split of B)
  std::vector<HTask> tasks[4];
  std::vector<HTile> tiles[4], pieces[4];     // pieces: the K-split tiles of symmetric products
  std::vector<HSymSlot> slots[4];
  psh::Arena ar(workspace, workspace_bytes);
  const size_t total_tiles = htiles(desc, count);
  for (int i = 0; i < count; ++i) {
    const ps_gemm_bf16_desc& d = desc[i];
    if (d.a_tiled == 2) return PS_EUNSUPPORTED;   // fragment-major planes: ps_fd_cy_step_f32 only
    if (!d.a_hi || !d.b_hi || !d.c || d.m < 1 || d.n < 1 || d.k < 1 ||
        (!d.a_tiled && d.lda < d.k) || d.ldb < d.k || d.ldc < d.n)
      return PS_EINVAL;
    // 16-byte loads of 8 consecutive k: k, both leading dimensions and the bases aligned
    if (d.k % HBK != 0 || (!d.a_tiled && d.lda % 8 != 0) || d.ldb % 8 != 0 || ((uintptr_t)d.a_hi % 16) != 0 ||
        ((uintptr_t)d.b_hi % 16) != 0 || (d.a_lo && ((uintptr_t)d.a_lo % 16) != 0) ||
        (d.b_lo && ((uintptr_t)d.b_lo % 16) != 0))
      return PS_EUNSUPPORTED;
    const int g = (d.a_lo ? 2 : 0) + (d.b_lo ? 1 : 0);
    const int tid = (int)tasks[g].size();
    HTask t{(const uint16_t*)d.a_hi, (const uint16_t*)d.a_lo, (const uint16_t*)d.b_hi,
            (const uint16_t*)d.b_lo, d.c, d.m, d.n, d.k, d.lda, d.ldb, d.ldc, 1, d.k, nullptr,
            d.a_tiled ? 1 : 0, d.symmetric ? 1 : 0, nullptr};
    if (d.symmetric) {
      // c = a a^T: same planes on both sides, square output
      if (d.m != d.n || d.a_tiled || d.b_hi != d.a_hi || d.b_lo != d.a_lo || d.ldb != d.lda)
        return PS_EINVAL;
      const int T = (d.m + TILE - 1) / TILE;
      const SymPlan sp = sym_plan(d.m, d.k);
      const int64_t nt = sp.tiles;
      const int tail = sp.tail, kc = sp.kc, nks = sp.nks;
      if (tail > 0) t.ptile = ar.take<float>((size_t)tail * nks * TILE * TILE);
      tasks[g].push_back(t);
      int64_t idx = 0;
      const int slot0 = (int)slots[g].size();
      auto emit = [&](int tm, int tn) {
        if (idx >= nt - tail) {                       // the last `tail` tiles: cut along k
          const int slot = (int)slots[g].size() - slot0;
          slots[g].push_back({tid, (short)tm, (short)tn, nks, slot});
          for (int ks = 0; ks < nks; ++ks) pieces[g].push_back({tid, (short)tm, (short)tn, ks, kc, nks, slot});
        } else {
          tiles[g].push_back({tid, (short)tm, (short)tn, 0, 0, 1, -1});
        }
        ++idx;
      };
      for (int tm = 0; tm < T; ++tm) emit(tm, tm);    // diagonal tiles first: never in the tail
      for (int tm = 0; tm < T; ++tm)
        for (int tn = tm + 1; tn < T; ++tn) emit(tm, tn);
      continue;
    }
    const int sp = hsplit_for(total_tiles, d.k);
    if (sp > 1) {
      t.kchunk = psh::round_up((d.k + sp - 1) / sp, HBK);
      t.ksplit = (d.k + t.kchunk - 1) / t.kchunk;
      if (t.ksplit > 1) t.partial = ar.take<float>((size_t)t.ksplit * d.m * d.n);
      else t.kchunk = d.k;
    }
    tasks[g].push_back(t);
    for (int tm = 0; tm < (d.m + TILE - 1) / TILE; ++tm)
      for (int tn = 0; tn < (d.n + TILE - 1) / TILE; ++tn)
        for (int ks = 0; ks < t.ksplit; ++ks) tiles[g].push_back({tid, (short)tm, (short)tn, ks, 0, 1, -1});
  }
  static std::once_flag attr_once;
  std::call_once(attr_once, [] {
    const int big = (int)((size_t)2 * 4 * HOP * sizeof(uint16_t));
    (void)hipFuncSetAttribute((const void*)gemm_bf16_grouped_kernel<1, 1>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_grouped_kernel<1, 2>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_grouped_kernel<2, 1>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_grouped_kernel<2, 2>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, big);
  });
  for (int g = 0; g < 4; ++g) {
    if (tasks[g].empty()) continue;
    if (!pieces[g].empty()) {
      // xcd_remap hands list chunk x (sizes q + 1, ..., q) to XCD x in order: every chunk gets its
      // contiguous share of the whole tiles first, then its share of the pieces, so that all XCDs
      // finish their whole tiles together and the pieces fill the last round everywhere
      const size_t F = tiles[g].size(), n = F + pieces[g].size();
      std::vector<HTile> all;
      all.reserve(n);
      size_t fi = 0, pi = 0;
      for (size_t x = 0; x < (size_t)psh::NXCD; ++x) {
        const size_t sx = n / psh::NXCD + (x < n % psh::NXCD), fx = F / psh::NXCD + (x < F % psh::NXCD);
        for (size_t i = 0; i < fx; ++i) all.push_back(tiles[g][fi++]);
        for (size_t i = fx; i < sx; ++i) all.push_back(pieces[g][pi++]);
      }
      tiles[g].swap(all);
    }
    HPlan::Group& gr = pl.g[g];
    gr.dt = ar.take<HTask>(tasks[g].size());
    gr.dl = ar.take<HTile>(tiles[g].size());
    if (ar.overflow) return PS_EWORKSPACE;
    PS_RC(psh::upload_async(st, gr.dt, tasks[g].data(), sizeof(HTask) * tasks[g].size()));
    PS_RC(psh::upload_async(st, gr.dl, tiles[g].data(), sizeof(HTile) * tiles[g].size()));
    gr.nt = (int)tiles[g].size();
    gr.ntasks = (int)tasks[g].size();
    for (auto& t : tasks[g]) { gr.any_split |= t.ksplit > 1; gr.any_sym |= t.sym != 0; }
    if (!slots[g].empty()) {
      gr.ds = ar.take<HSymSlot>(slots[g].size());
      if (ar.overflow) return PS_EWORKSPACE;
      PS_RC(psh::upload_async(st, gr.ds, slots[g].data(), sizeof(HSymSlot) * slots[g].size()));
      gr.nslots = (int)slots[g].size();
    }
  }
  return PS_OK;
}

static int hplan_launch(hipStream_t st, const HPlan& pl) {
  for (int g = 0; g < 4; ++g) {
    const HPlan::Group& gr = pl.g[g];
    if (gr.nt == 0) continue;
    const int sa = (g & 2) ? 2 : 1, sb = (g & 1) ? 2 : 1;
    size_t lds = (size_t)2 * (sa + sb) * HOP * sizeof(uint16_t);  // 40 .. 80 KiB
    if (gr.any_sym) lds = std::max(lds, (size_t)TILE * SYM_TLD * sizeof(float));   // mirror staging
    if (g == 0)
      hipLaunchKernelGGL((gemm_bf16_grouped_kernel<1, 1>), dim3(gr.nt), dim3(256), lds, st, gr.dt, gr.dl, gr.nt);
    else if (g == 1)
      hipLaunchKernelGGL((gemm_bf16_grouped_kernel<1, 2>), dim3(gr.nt), dim3(256), lds, st, gr.dt, gr.dl, gr.nt);
    else if (g == 2)
      hipLaunchKernelGGL((gemm_bf16_grouped_kernel<2, 1>), dim3(gr.nt), dim3(256), lds, st, gr.dt, gr.dl, gr.nt);
    else
      hipLaunchKernelGGL((gemm_bf16_grouped_kernel<2, 2>), dim3(gr.nt), dim3(256), lds, st, gr.dt, gr.dl, gr.nt);
    if (gr.any_split)
      hipLaunchKernelGGL(gemm_bf16_splitk_reduce_kernel, dim3((unsigned)gr.ntasks, 256), dim3(256), 0,
                         st, gr.dt);
    if (gr.nslots > 0)
      hipLaunchKernelGGL(gemm_bf16_symtile_reduce_kernel, dim3((unsigned)gr.nslots * SYM_RED), dim3(256),
                         0, st, gr.dt, gr.ds);
    PS_LAUNCH_CHECK();
  }
  return PS_OK;
}

extern "C" int ps_gemm_bf16_grouped(void* stream, const ps_gemm_bf16_desc* desc, int count,
                                    void* workspace, size_t workspace_bytes) {
  PS_DEVICE_CHECK();
  HPlan pl;
  PS_RC(hplan_build((hipStream_t)stream, desc, count, workspace, workspace_bytes, pl));
  return hplan_launch((hipStream_t)stream, pl);
}

// At most 128 blocks of 64 rows (one 4096-dim factor is 64): the fused kernels run on 32-row blocks, i.e.
// on at most one workgroup per CU
#ifndef PS_FD_HALF_BELOW
#define PS_FD_HALF_BELOW 129   // measured at n = 4096, b = 96: 1 / 2 factors 35 / 40 us (64-row: 48 / 51), 3 factors 73 (56)
#endif
static bool fd_half_blocks(int batch, int64_t n) { return (int64_t)batch * (n / 64) < PS_FD_HALF_BELOW; }

extern "C" int ps_fd_cy_step_f32(void* stream, const void* const* c_hi, const void* const* c_lo,
                                 int batch, const void* yt_hi, const void* yt_lo, const float* y,
                                 const float* y_prev, float* y_next, void* nt_hi, void* nt_lo,
                                 const float* params, int step, int64_t n, int64_t b) {
  PS_DEVICE_CHECK();
  if (!c_hi || !c_lo || batch < 1 || !yt_hi || !yt_lo || !y || !y_prev || !y_next || !params ||
      step < 2 || (nt_hi == nullptr) != (nt_lo == nullptr) || nt_hi == yt_hi || nt_lo == yt_lo)
    return PS_EINVAL;
  if (batch > 16 || n < 128 || n % 128 != 0 || b % 32 != 0 || b < 32 || b > 96 ||
      (int64_t)batch * (n / 64) > 0x7fffffff)
    return PS_EUNSUPPORTED;
  FdCyArgs a{};
  for (int j = 0; j < batch; ++j) {
    if (!c_hi[j] || !c_lo[j] || ((uintptr_t)c_hi[j] % 16) != 0 || ((uintptr_t)c_lo[j] % 16) != 0)
      return PS_EINVAL;
    a.a_hi[j] = (const uint16_t*)c_hi[j];
    a.a_lo[j] = (const uint16_t*)c_lo[j];
  }
  if (((uintptr_t)yt_hi % 16) != 0 || ((uintptr_t)yt_lo % 16) != 0 || ((uintptr_t)y % 16) != 0 ||
      ((uintptr_t)y_prev % 16) != 0 || ((uintptr_t)y_next % 16) != 0 ||
      (nt_hi && (((uintptr_t)nt_hi % 16) != 0 || ((uintptr_t)nt_lo % 16) != 0)))
    return PS_EUNSUPPORTED;
  a.bt_hi = (const uint16_t*)yt_hi; a.bt_lo = (const uint16_t*)yt_lo;
  a.y = y; a.y_prev = y_prev; a.y_next = y_next;
  a.nt_hi = (uint16_t*)nt_hi; a.nt_lo = (uint16_t*)nt_lo;
  a.params = params; a.step = step; a.n = (int)n;
  hipStream_t st = (hipStream_t)stream;
  if (fd_half_blocks(batch, n)) {   // one or two factors: 32-row workgroups, twice as many
    a.nwg = (int)(batch * (n / 32));
    if (b == 96) hipLaunchKernelGGL((fd_cy_step_kernel<3, 1>), dim3(a.nwg), dim3(256), 0, st, a);
    else if (b == 64) hipLaunchKernelGGL((fd_cy_step_kernel<2, 1>), dim3(a.nwg), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((fd_cy_step_kernel<1, 1>), dim3(a.nwg), dim3(256), 0, st, a);
  } else {
    a.nwg = (int)(batch * (n / 64));
    if (b == 96) hipLaunchKernelGGL((fd_cy_step_kernel<3>), dim3(a.nwg), dim3(256), 0, st, a);
    else if (b == 64) hipLaunchKernelGGL((fd_cy_step_kernel<2>), dim3(a.nwg), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((fd_cy_step_kernel<1>), dim3(a.nwg), dim3(256), 0, st, a);
  }
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_convert_f32_to_bf16x3_frag(void* stream, const float* src, void* p0, void* p1, void* p2,
                                             int64_t rows, int64_t cols, int64_t lds) {
  PS_DEVICE_CHECK();
  if (!src || !p0 || !p1 || !p2 || rows < 1 || cols < 1 || lds < cols) return PS_EINVAL;
  if (rows % 64 != 0 || cols % 64 != 0) return PS_EUNSUPPORTED;
  const int64_t tr = rows / 64, tc = cols / 64;
  if (tr * tc > 0x7fffffff) return PS_EUNSUPPORTED;
  hipLaunchKernelGGL(cvt_bf16_kernel, dim3((unsigned)(tr * tc)), dim3(256), 0, (hipStream_t)stream, src,
                     (uint16_t*)p0, (uint16_t*)p1, (int)rows, (int)cols, lds, cols, 3, (int)tc,
                     (uint16_t*)p2);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_fd_cx6_f32(void* stream, const void* const* c0, const void* const* c1,
                             const void* const* c2, int batch, const float* x, float* z, void* xt0,
                             void* xt1, void* xt2, int64_t n, int64_t b) {
  PS_DEVICE_CHECK();
  if (!c0 || !c1 || !c2 || batch < 1 || !x || !z || !xt0 || !xt1 || !xt2 || x == z) return PS_EINVAL;
  if (batch > 16 || n < 128 || n % 128 != 0 || b % 32 != 0 || b < 32 || b > 96 ||
      (int64_t)batch * (n / 64) > 0x7fffffff)
    return PS_EUNSUPPORTED;
  FdCx6Args a{};
  for (int j = 0; j < batch; ++j) {
    const void* p[3] = {c0[j], c1[j], c2[j]};
    for (int pl = 0; pl < 3; ++pl) {
      if (!p[pl] || ((uintptr_t)p[pl] % 16) != 0) return PS_EINVAL;
      a.a[pl][j] = (const uint16_t*)p[pl];
    }
  }
  if (((uintptr_t)x % 16) != 0 || ((uintptr_t)z % 16) != 0 || ((uintptr_t)xt0 % 16) != 0 ||
      ((uintptr_t)xt1 % 16) != 0 || ((uintptr_t)xt2 % 16) != 0)
    return PS_EUNSUPPORTED;
  a.bt[0] = (const uint16_t*)xt0; a.bt[1] = (const uint16_t*)xt1; a.bt[2] = (const uint16_t*)xt2;
  a.z = z; a.n = (int)n;
  hipStream_t st = (hipStream_t)stream;
  const bool half = fd_half_blocks(batch, n);
  a.nwg = (int)(batch * (n / (half ? 32 : 64)));
  const dim3 sgrid((unsigned)(batch * (n / 64))), grid(a.nwg), blk(256);
#define PS_CX6(CBV)                                                                                         \
  do {                                                                                                      \
    hipLaunchKernelGGL((fd_split3_frag_kernel<CBV>), sgrid, blk, 0, st, x, (uint16_t*)xt0, (uint16_t*)xt1,  \
                       (uint16_t*)xt2, (int)n);                                                             \
    if (half) hipLaunchKernelGGL((fd_cx6_kernel<CBV, 1>), grid, blk, 0, st, a);                             \
    else hipLaunchKernelGGL((fd_cx6_kernel<CBV, 2>), grid, blk, 0, st, a);                                  \
  } while (0)
  if (b == 96) PS_CX6(3);
  else if (b == 64) PS_CX6(2);
  else PS_CX6(1);
#undef PS_CX6
  PS_LAUNCH_CHECK();
  return PS_OK;
}

// One whole Chebyshev filter of the subspace iteration (precondition_amd/subspace.py) in one call:
//   step 1:            y1 = recurrence(z, y0)                           (z = C y0 is given)
//   step s = 2..deg:   z = C y_{s-1}  (grouped bf16 product, plan built once)
//                      y_s = recurrence(z, y_{s-1}, y_{s-2})
// desc[j] describes z_j = C_j * yt_j^T with b_hi / b_lo pointing into the transposed bf16 copies
// (yt_hi / yt_lo, leading dimension ldt) that every recurrence step rewrites; y0, y1, y2 are the
// three rotating [batch][n][b] float32 iterates.  *result_index receives which of them holds the
// filtered block.  ~3 launches per step and no host work between them (the Python loop spent ~40 us
// per launch building descriptor tables: 10 of the 11 ms of a one-factor update).
extern "C" int ps_fd_filter_round_f32(void* stream, const ps_gemm_bf16_desc* desc, int batch,
                                      float* z, float* y0, float* y1, float* y2, void* yt_hi,
                                      void* yt_lo, const float* params, int max_degree, int64_t n,
                                      int64_t b, int64_t ldt, void* workspace,
                                      size_t workspace_bytes, int32_t* result_index) {
  PS_DEVICE_CHECK();
  if (!desc || batch <= 0 || !z || !y0 || !y1 || !y2 || !params || max_degree < 1 || !result_index ||
      (max_degree >= 2 && (!yt_hi || !workspace)))
    return PS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool more = max_degree >= 2;
  int frag = 0;
  for (int j = 0; j < batch; ++j) frag += desc[j].a_tiled == 2;
  if (frag != 0) {
    // Fragment-major covariances: every step after the first is ONE launch (fd_cy_step_kernel:
    // product + recurrence + the next operand); yt_hi / yt_lo hold TWO copies of the iterate
    // planes (2 * batch * n * b elements each), written and read alternately.
    if (frag != batch) return PS_EINVAL;
    if (more && !yt_lo) return PS_EUNSUPPORTED;
    const void* ch[16]; const void* cl[16];
    if (batch > 16) return PS_EUNSUPPORTED;
    for (int j = 0; j < batch; ++j) {
      if (desc[j].m != n || desc[j].k != n || desc[j].n != b) return PS_EINVAL;
      ch[j] = desc[j].a_hi; cl[j] = desc[j].a_lo;
    }
    const int64_t plane = (int64_t)batch * n * b;
    uint16_t* th[2] = {(uint16_t*)yt_hi, (uint16_t*)yt_hi + plane};
    uint16_t* tl[2] = {(uint16_t*)yt_lo, (uint16_t*)yt_lo + plane};
    PS_RC(ps_fd_filter_step_f32(stream, z, y0, nullptr, y1, more ? th[0] : nullptr,
                                more ? tl[0] : nullptr, params, 1, batch, n, b, 0));
    float* bufs[3] = {y0, y1, y2};
    int ip = 0, iy = 1, in = 2, cur = 0;
    for (int step = 2; step <= max_degree; ++step) {
      const bool again = step < max_degree;
      PS_RC(ps_fd_cy_step_f32(stream, ch, cl, batch, th[cur], tl[cur], bufs[iy], bufs[ip], bufs[in],
                              again ? th[cur ^ 1] : nullptr, again ? tl[cur ^ 1] : nullptr, params,
                              step, n, b));
      const int t = ip; ip = iy; iy = in; in = t;
      cur ^= 1;
    }
    *result_index = iy;
    return PS_OK;
  }
  PS_RC(ps_fd_filter_step_f32(stream, z, y0, nullptr, y1, more ? yt_hi : nullptr,
                              more ? yt_lo : nullptr, params, 1, batch, n, b, ldt));
  float* bufs[3] = {y0, y1, y2};
  int ip = 0, iy = 1, in = 2;
  if (more) {
    HPlan pl;
    PS_RC(hplan_build(st, desc, batch, workspace, workspace_bytes, pl));
    for (int step = 2; step <= max_degree; ++step) {
      PS_RC(hplan_launch(st, pl));
      const bool again = step < max_degree;
      PS_RC(ps_fd_filter_step_f32(stream, z, bufs[iy], bufs[ip], bufs[in], again ? yt_hi : nullptr,
                                  again ? yt_lo : nullptr, params, step, batch, n, b, ldt));
      const int t = ip; ip = iy; iy = in; in = t;
    }
  }
  *result_index = iy;
  return PS_OK;
}
