// gemm_core.hip.h — 128x128 output-tile float32 GEMM core for gfx950 (MI355X).
//
// One workgroup = 256 threads = 4 wavefronts (64 lanes each) arranged 2x2; each
// wavefront owns a 64x64 sub-tile as 2x2 accumulators of
// v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain, 64 cycles/SIMD).
// Operands are staged global -> registers -> LDS (double buffered, one barrier
// per K-tile); the global loads for K-tile t+1 are issued before the MFMAs of
// tile t and written to the other LDS buffer after them.
//
// Two LDS images, chosen by how the operand lies in memory so that staging never
// transposes:
//   KC ("k-contiguous", element (mn,k) at p[mn*ld + k]):  LDS [128][BK+4];
//       a lane reads 4 consecutive k with one ds_read_b128 (conflict-free: row
//       stride 20 or 36 dwords walks all 64 banks in a 16-lane group).
//   MC ("mn-contiguous", element (mn,k) at p[k*ld + mn]):  LDS [BK][128+4];
//       a lane reads one dword per k (consecutive lanes, consecutive banks).
// Inside every 8-deep k-chunk the k order is permuted (lane half h consumes
// k = 8c+4h+s at MFMA step s) identically for A and B, which only changes the
// summation order of the exact-f32 chain.
//
// Replaces the XLA dot at DS:671,674,845,846 (Newton products) and DS:1469
// (statistics Gram matrix) of the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TILE = 128;      // BM = BN
constexpr int NTHREADS = 256;  // 4 wavefronts

enum Layout { KC = 0, MC = 1 };

// Pointers that come out of descriptor tables in memory are generic ("flat") to
// the compiler; flat loads count on lgkmcnt as well as vmcnt and so would
// serialise with the LDS fragment reads.  Everything we touch through such
// pointers is device global memory: say so.
#define PS_GLOBAL __attribute__((address_space(1)))
__device__ inline f32x4 gload4(const float* p) {
  return *(const f32x4 PS_GLOBAL*)(p);
}
__device__ inline float gload1(const float* p) { return *(const float PS_GLOBAL*)(p); }
__device__ inline void gstore1(float* p, float v) { *(float PS_GLOBAL*)(p) = v; }
// Write-through (sc1) store: the bytes leave this XCD's L2 at once, so a workgroup on
// another CU / XCD can be handed the data inside a launch after the storing waves have
// drained (s_waitcnt vmcnt(0)) and one lane has signalled (cdna_hip_programming.md G16 R1).
__device__ inline void gstore1_wt(float* p, float v) {
  __hip_atomic_store((float PS_GLOBAL*)(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool WT>
__device__ inline void tstore1(float* p, float v) {
  if (WT) gstore1_wt(p, v); else gstore1(p, v);
}

template <int BK>
struct SmemCfg {
  static constexpr int KC_LD = BK + 4;
  static constexpr int MC_LD = TILE + 4;
  static constexpr int KC_SIZE = TILE * KC_LD;
  static constexpr int MC_SIZE = BK * MC_LD;
  static constexpr int OP_SIZE = KC_SIZE > MC_SIZE ? KC_SIZE : MC_SIZE;
  static constexpr int STAGE_SIZE = 2 * OP_SIZE;    // A + B
  static constexpr int TOTAL = 2 * STAGE_SIZE;      // double buffered (floats): any layout pair
  // exact footprint of one layout pair (the K loop packs A and B images back to back)
  static constexpr int op_size(int layout) { return layout == 0 ? KC_SIZE : MC_SIZE; }
  static constexpr int total(int la, int lb) { return 2 * (op_size(la) + op_size(lb)); }
};

struct Operand {
  const float* p;
  int ld;
  int mn0;   // first row (A) / column (B) of this tile
  int MN;    // logical extent along mn (guarded loads only)
  int K;     // logical extent along k  (guarded loads only)
  bool vec;  // 16-byte vector loads are legal (base and ld aligned)
};

// ---- global -> registers ---------------------------------------------------
template <int LAYOUT, int BK, bool GUARD>
struct TileLoader {
  static constexpr int NV = (TILE * BK / 4) / NTHREADS;  // float4 per thread
  static_assert(NV >= 1, "BK too small");

  __device__ static inline void coords(int v, int tid, int& mn, int& k) {
    const int f = tid + NTHREADS * v;
    if (LAYOUT == KC) {
      mn = f / (BK / 4);
      k = (f % (BK / 4)) * 4;
    } else {
      k = f / (TILE / 4);
      mn = (f % (TILE / 4)) * 4;
    }
  }

  // Unguarded loads address memory as (wave-uniform 64-bit base) + (one 32-bit byte offset per
  // lane): base = tile origin advanced to K-tile k0 and to the v-th group of rows, offset = the
  // lane's position inside the first group.  The uniform part lives in SGPRs (global_load ...
  // saddr form), so a tile costs ONE address VGPR per operand instead of a 64-bit pair per load:
  // with two register sets in flight that is the difference between 256 VGPRs + scratch and none.
  // The offset spans at most ROWS_PER_V rows of the operand, far below 2^32 bytes.
  static constexpr int ROWS_PER_V = LAYOUT == KC ? NTHREADS / (BK / 4) : NTHREADS / (TILE / 4);
  __device__ static inline uint32_t lane_byte_offset(int ld, int tid) {
    int mn, k;
    coords(0, tid, mn, k);
    return 4u * (LAYOUT == KC ? (uint32_t)mn * (uint32_t)ld + (uint32_t)k
                              : (uint32_t)k * (uint32_t)ld + (uint32_t)mn);
  }

  __device__ static inline void load(const Operand& op, int k0, int tid,
                                     f32x4 (&r)[NV]) {
    if (!GUARD) {
      uint32_t off = lane_byte_offset(op.ld, tid);
      // opaque per call: keeps the zero-extension of the offset next to the loads (instruction
      // selection sees "SGPR base + zext(VGPR)" and emits the saddr form; hoisted out of the K
      // loop it becomes a 64-bit VGPR add per load, and VALU work costs fp32-MFMA cycles)
      asm volatile("" : "+v"(off));
      const float* base = LAYOUT == KC ? op.p + (int64_t)op.mn0 * op.ld + k0
                                       : op.p + (int64_t)k0 * op.ld + op.mn0;
#pragma unroll
      for (int v = 0; v < NV; ++v)
        r[v] = *(const f32x4 PS_GLOBAL*)((const char PS_GLOBAL*)(base + (int64_t)(v * ROWS_PER_V) * op.ld) +
                                         (uint64_t)off);
      return;
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int mn, k;
      coords(v, tid, mn, k);
      const int gmn = op.mn0 + mn;
      const int gk = k0 + k;
      if (LAYOUT == KC) {
        const float* src = op.p + (int64_t)gmn * op.ld + gk;
        if (!GUARD) {
          r[v] = gload4(src);
        } else {
          f32x4 t = {0.f, 0.f, 0.f, 0.f};
          if (gmn < op.MN) {
            if (op.vec && gk + 3 < op.K) {
              t = gload4(src);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (gk + e < op.K) t[e] = gload1(src + e);
            }
          }
          r[v] = t;
        }
      } else {
        const float* src = op.p + (int64_t)gk * op.ld + gmn;
        if (!GUARD) {
          r[v] = gload4(src);
        } else {
          f32x4 t = {0.f, 0.f, 0.f, 0.f};
          if (gk < op.K) {
            if (op.vec && gmn + 3 < op.MN) {
              t = gload4(src);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (gmn + e < op.MN) t[e] = gload1(src + e);
            }
          }
          r[v] = t;
        }
      }
    }
  }

  // registers -> LDS image
  __device__ static inline void store(float* s, int tid, const f32x4 (&r)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int mn, k;
      coords(v, tid, mn, k);
      float* dst = (LAYOUT == KC) ? s + mn * SmemCfg<BK>::KC_LD + k
                                  : s + k * SmemCfg<BK>::MC_LD + mn;
      *reinterpret_cast<f32x4*>(dst) = r[v];
    }
  }
};

// ---- LDS -> fragments -> MFMA for one K-tile --------------------------------
template <int LA, int LB, int BK>
__device__ inline void compute_ktile(const float* sA, const float* sB,
                                     f32x16 (&acc)[2][2], int wm, int wn,
                                     int lane) {
  const int i = lane & 31;
  const int h = lane >> 5;
#pragma unroll
  for (int c = 0; c < BK / 8; ++c) {
    float a[2][4], b[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (LA == KC) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(
            sA + (wm * 64 + t * 32 + i) * SmemCfg<BK>::KC_LD + 8 * c + 4 * h);
        a[t][0] = v[0]; a[t][1] = v[1]; a[t][2] = v[2]; a[t][3] = v[3];
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          a[t][s] = sA[(8 * c + 4 * h + s) * SmemCfg<BK>::MC_LD + wm * 64 + t * 32 + i];
      }
      if (LB == KC) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(
            sB + (wn * 64 + t * 32 + i) * SmemCfg<BK>::KC_LD + 8 * c + 4 * h);
        b[t][0] = v[0]; b[t][1] = v[1]; b[t][2] = v[2]; b[t][3] = v[3];
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          b[t][s] = sB[(8 * c + 4 * h + s) * SmemCfg<BK>::MC_LD + wn * 64 + t * 32 + i];
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
              a[tm][s], b[tn][s], acc[tm][tn], 0, 0, 0);
    }
  }
}

// ---- whole K loop for one 128x128 tile --------------------------------------
// A supplies rows (mn = m), B supplies columns (mn = n).  Kext is the k range to
// cover (rounded up to BK internally; unguarded callers must have zero padding
// there).  smem must hold SmemCfg<BK>::TOTAL floats.  Ends with a barrier, so
// smem may be reused by the caller's epilogue.
__device__ inline void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
}


// ---- segmented accumulation (ps_options.accumulation = PS_ACCUM_SEGMENTED) ------------------
// The fp32 MFMA is one fmaf chain over k per output element: its rounding error grows like
// sqrt(K) * eps relative to sum |a||b|.  With `seg` set the chain is cut every SEG_K = 128 values of
// k: the finished segment is added to a second register set (`total`) and the chain restarts from
// zero -- blocked summation, error ~ (sqrt(SEG_K) + sqrt(K / SEG_K)) * eps.  Measured on the cond
// ~3.5e3 (n = 512) and ~7e3 (n = 1024) p = 4 blocks of the ViT-B state (tools/dev_chain_accuracy.py,
// bit-level emulation of the chains): root error vs float64 1.30x NumPy/OpenBLAS's with one chain,
// 1.00x (to the last digit) with segments of 256 -- OpenBLAS's own K blocking on the build box --
// and 0.75x with 128.  64 v_add per segment and 64 more registers; same MFMA work.  Only the
// SEGV instantiations carry `total`: the default product kernel (two register sets of loads, two
// workgroups per CU, 220 VGPRs) has no room for it, so the careful kernel runs ONE workgroup per CU
// with the whole register file (a lone workgroup keeps the fp32 MFMA pipe ~94 % as busy as two).
constexpr int SEG_K = 128;
__device__ __forceinline__ void seg_flush(f32x16 (&acc)[2][2], f32x16 (&total)[2][2]) {
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        total[tm][tn][r] = __fadd_rn(total[tm][tn][r], acc[tm][tn][r]);
        acc[tm][tn][r] = 0.f;
      }
}
__device__ __forceinline__ void seg_add(const f32x16 (&acc)[2][2], f32x16 (&total)[2][2]) {
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) total[tm][tn][r] = __fadd_rn(total[tm][tn][r], acc[tm][tn][r]);
}
__device__ __forceinline__ void seg_finish(f32x16 (&acc)[2][2], const f32x16 (&total)[2][2]) {
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = __fadd_rn(total[tm][tn][r], acc[tm][tn][r]);
}

// The two register sets of the DEEP K loop, split in two calls so that a caller that walks a
// list of tiles can issue the first loads of its NEXT tile before the epilogue of the current
// one (deep_issue_first), and enter that tile's K loop with them in flight (deep_run).
template <int LA, int LB, int BK, bool GUARD>
struct DeepSets {
  f32x4 ra0[TileLoader<LA, BK, GUARD>::NV], rb0[TileLoader<LB, BK, GUARD>::NV];
  f32x4 ra1[TileLoader<LA, BK, GUARD>::NV], rb1[TileLoader<LB, BK, GUARD>::NV];
};

// Issues the global loads of K-tiles 0 and 1 (set 0 and set 1).
template <int LA, int LB, int BK, bool GUARD>
__device__ inline void deep_issue_first(const Operand& A, const Operand& B, int Kext,
                                        DeepSets<LA, LB, BK, GUARD>& r) {
  using LdA = TileLoader<LA, BK, GUARD>;
  using LdB = TileLoader<LB, BK, GUARD>;
  const int tid = threadIdx.x;
  const int nk = (Kext + BK - 1) / BK;
  LdA::load(A, 0, tid, r.ra0);
  LdB::load(B, 0, tid, r.rb0);
  // unconditional (see deep_run): a one-K-tile product re-reads K-tile 0 into the unused set
  const int k1 = nk > 1 ? BK : 0;
  LdA::load(A, k1, tid, r.ra1);
  LdB::load(B, k1, tid, r.rb1);
}

// acc += A * B with K-tiles 0 and 1 already requested by deep_issue_first.  Ends with a barrier.
template <int LA, int LB, int BK, bool GUARD, bool SEGV = false>
__device__ inline void deep_run(const Operand& A, const Operand& B, int Kext, float* smem,
                                f32x16 (&acc)[2][2], DeepSets<LA, LB, BK, GUARD>& r,
                                unsigned long long* t_fill = nullptr, bool seg = false) {
  using LdA = TileLoader<LA, BK, GUARD>;
  using LdB = TileLoader<LB, BK, GUARD>;
  constexpr int OPS = SmemCfg<BK>::op_size(LA);
  constexpr int STG = SmemCfg<BK>::op_size(LA) + SmemCfg<BK>::op_size(LB);
  constexpr int SEGT = SEG_K / BK;   // K-tiles per segment (even)
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int wm = wave >> 1;
  const int wn = wave & 1;
  const int nk = (Kext + BK - 1) / BK;
  f32x16 total[SEGV ? 2 : 1][SEGV ? 2 : 1];
  if constexpr (SEGV) zero_acc(total);
  LdA::store(smem, tid, r.ra0);
  LdB::store(smem + OPS, tid, r.rb0);
  __syncthreads();
  if (t_fill != nullptr && tid == 0) *t_fill = __builtin_amdgcn_s_memrealtime();  // dev trace
  float* s0 = smem;
  float* s1 = smem + STG;
  for (int kt = 0; kt < nk; kt += 2) {
    // even iteration: tile kt in s0; set 1 holds tile kt+1; set 0 is free -> tile kt+2
    // The loads are UNCONDITIONAL: the vmcnt the compiler puts in front of the LDS writes of the
    // other set must hold on every path, and with a skipped load on one path it has to assume
    // that the set's loads are the youngest ones in flight -- every K-tile then waited for the
    // loads issued one K-tile (not two) earlier.  Past the end of the product the request
    // re-reads the last K-tile (a cache hit that nobody uses).
    LdA::load(A, min(kt + 2, nk - 1) * BK, tid, r.ra0);
    LdB::load(B, min(kt + 2, nk - 1) * BK, tid, r.rb0);
    compute_ktile<LA, LB, BK>(s0, s0 + OPS, acc, wm, wn, lane);
    if (kt + 1 < nk) {
      LdA::store(s1, tid, r.ra1);
      LdB::store(s1 + OPS, tid, r.rb1);
    }
    __syncthreads();
    if (kt + 1 >= nk) break;
    // odd iteration: tile kt+1 in s1; set 0 holds tile kt+2; set 1 is free -> tile kt+3
    LdA::load(A, min(kt + 3, nk - 1) * BK, tid, r.ra1);
    LdB::load(B, min(kt + 3, nk - 1) * BK, tid, r.rb1);
    compute_ktile<LA, LB, BK>(s1, s1 + OPS, acc, wm, wn, lane);
    if (kt + 2 < nk) {
      LdA::store(s0, tid, r.ra0);
      LdB::store(s0 + OPS, tid, r.rb0);
    }
    __syncthreads();
    if constexpr (SEGV) { if (seg && (kt + 2) % SEGT == 0 && kt + 2 < nk) seg_flush(acc, total); }
  }
  if constexpr (SEGV) { if (seg) seg_finish(acc, total); }
  (void)total; (void)SEGT;
}

// ---- software-pipelined K loop (any operand layouts, BK = 32, unguarded) --------------------
// deep_run's K-tile body lets the compiler place the fragment reads: it issues each ds_read just
// before its first use (minimal registers), so a wavefront that is alone on its SIMD stalls on
// LDS latency several times per 16-MFMA chunk and once more, with a drained MFMA pipe, at the
// barrier that ends every K-tile (tools/dev_stage_trace.py: a workgroup that is alone on its CU
// runs its K loop at 0.58 of the MFMA rate, two at 0.8).  Here the schedule is explicit:
//   * fragments are double buffered by 8-deep chunk: the ds_reads of chunk c+1 are issued between
//     the MFMA groups of chunk c (sched_barrier fences keep the order);
//   * the next K-tile's LDS image is written during chunk 1 and the workgroup barrier sits
//     BEFORE chunk 3, whose MFMAs then cover the reads of chunk 0 of the next K-tile: no drained
//     pipe at the K-tile boundary.  Safe with two LDS stages: every read of a stage is issued
//     before the barrier of its K-tile (and waited for by it), and the stage is next written
//     after that barrier.
// Same MFMA order per accumulator as compute_ktile: bit-identical results.  The global loads of
// K-tile t+2 are issued at the top of K-tile t and consumed during chunk 1 of K-tile t+1.
struct FragKM {
  float a[2][4], b[2][4];
};

template <int BK, int LA = KC>
__device__ __forceinline__ void pipe_read_a(const float* sA, int c, int wm, int i, int h, FragKM& f) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (LA == KC) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(
          sA + (wm * 64 + t * 32 + i) * SmemCfg<BK>::KC_LD + 8 * c + 4 * h);
      f.a[t][0] = v[0]; f.a[t][1] = v[1]; f.a[t][2] = v[2]; f.a[t][3] = v[3];
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        f.a[t][s] = sA[(8 * c + 4 * h + s) * SmemCfg<BK>::MC_LD + wm * 64 + t * 32 + i];
    }
  }
}
template <int BK>
__device__ __forceinline__ void pipe_read_b(const float* sB, int c, int s, int wn, int i, int h,
                                            FragKM& f) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
    f.b[t][s] = sB[(8 * c + 4 * h + s) * SmemCfg<BK>::MC_LD + wn * 64 + t * 32 + i];
}
// B image in the KC layout (rows = n): one 16-byte read per 32-column block, like A
template <int BK>
__device__ __forceinline__ void pipe_read_b_kc(const float* sB, int c, int wn, int i, int h, FragKM& f) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(
        sB + (wn * 64 + t * 32 + i) * SmemCfg<BK>::KC_LD + 8 * c + 4 * h);
    f.b[t][0] = v[0]; f.b[t][1] = v[1]; f.b[t][2] = v[2]; f.b[t][3] = v[3];
  }
}
// ZERO_C: the chain restarts here -- the MFMA takes the constant 0 as its C operand (an inline
// constant: no instruction is spent on clearing the accumulators of a new segment).
template <bool ZERO_C = false>
__device__ __forceinline__ void pipe_mfma(const FragKM& f, int s, f32x16 (&acc)[2][2]) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[tm][s], f.b[tn][s],
                                                         ZERO_C ? zero : acc[tm][tn], 0, 0, 0);
}
#define PS_FENCE() __builtin_amdgcn_sched_barrier(0)

// One chunk: the 16 MFMAs on `use`, with the reads of chunk (cn) of image (nA, nB) into `nxt`
// spread between the MFMA groups (rd = false: nothing to read).
template <int BK, int LA, int LB, bool ZERO_C = false>
__device__ __forceinline__ void pipe_chunk(const FragKM& use, FragKM& nxt, const float* nA,
                                           const float* nB, int cn, bool rd, int wm, int wn, int i,
                                           int h, f32x16 (&acc)[2][2]) {
  pipe_mfma<ZERO_C>(use, 0, acc);
  PS_FENCE();
  if (rd) {
    pipe_read_a<BK, LA>(nA, cn, wm, i, h, nxt);
    if (LB == MC) pipe_read_b<BK>(nB, cn, 0, wn, i, h, nxt);
  }
  PS_FENCE();
  pipe_mfma(use, 1, acc);
  PS_FENCE();
  if (rd) {
    if (LB == MC) {
      pipe_read_b<BK>(nB, cn, 1, wn, i, h, nxt); pipe_read_b<BK>(nB, cn, 2, wn, i, h, nxt);
      pipe_read_b<BK>(nB, cn, 3, wn, i, h, nxt);
    } else {
      pipe_read_b_kc<BK>(nB, cn, wn, i, h, nxt);
    }
  }
  PS_FENCE();
  pipe_mfma(use, 2, acc);
  pipe_mfma(use, 3, acc);
  PS_FENCE();
}

template <int BK, int LA, int LB>
__device__ inline void deep_run_pipe(const Operand& A, const Operand& B, int Kext, float* smem,
                                     f32x16 (&acc)[2][2], DeepSets<LA, LB, BK, false>& r,
                                     unsigned long long* t_fill = nullptr) {
  static_assert(BK == 32, "four 8-deep chunks per K-tile");
  using LdA = TileLoader<LA, BK, false>;
  using LdB = TileLoader<LB, BK, false>;
  constexpr int OPS = SmemCfg<BK>::op_size(LA);
  constexpr int STG = SmemCfg<BK>::op_size(LA) + SmemCfg<BK>::op_size(LB);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 31, h = lane >> 5;
  const int nk = (Kext + BK - 1) / BK;
  float* s0 = smem;
  float* s1 = smem + STG;
  LdA::store(s0, tid, r.ra0);
  LdB::store(s0 + OPS, tid, r.rb0);
  __syncthreads();
  if (t_fill != nullptr && tid == 0) *t_fill = __builtin_amdgcn_s_memrealtime();  // dev trace
  FragKM f0, f1;
  pipe_read_a<BK, LA>(s0, 0, wm, i, h, f0);
  if (LB == MC) {
#pragma unroll
    for (int s = 0; s < 4; ++s) pipe_read_b<BK>(s0 + OPS, 0, s, wn, i, h, f0);
  } else {
    pipe_read_b_kc<BK>(s0 + OPS, 0, wn, i, h, f0);
  }
  for (int kt = 0; kt < nk; kt += 2) {
    // ---- even K-tile: image in s0; set 1 holds K-tile kt+1 (-> s1); set 0 is free -> kt+2 ----
    {
      const bool nxt = kt + 1 < nk;
      LdA::load(A, min(kt + 2, nk - 1) * BK, tid, r.ra0);   // unconditional: see deep_run
      LdB::load(B, min(kt + 2, nk - 1) * BK, tid, r.rb0);
      PS_FENCE();
      pipe_chunk<BK, LA, LB>(f0, f1, s0, s0 + OPS, 1, true, wm, wn, i, h, acc);
      if (nxt) { LdA::store(s1, tid, r.ra1); LdB::store(s1 + OPS, tid, r.rb1); }
      PS_FENCE();
      pipe_chunk<BK, LA, LB>(f1, f0, s0, s0 + OPS, 2, true, wm, wn, i, h, acc);
      pipe_chunk<BK, LA, LB>(f0, f1, s0, s0 + OPS, 3, true, wm, wn, i, h, acc);
      __syncthreads();
      PS_FENCE();
      pipe_chunk<BK, LA, LB>(f1, f0, s1, s1 + OPS, 0, nxt, wm, wn, i, h, acc);
      if (!nxt) break;
    }
    // ---- odd K-tile: image in s1; set 0 holds K-tile kt+2 (-> s0); set 1 is free -> kt+3 ----
    {
      const bool nxt = kt + 2 < nk;
      LdA::load(A, min(kt + 3, nk - 1) * BK, tid, r.ra1);
      LdB::load(B, min(kt + 3, nk - 1) * BK, tid, r.rb1);
      PS_FENCE();
      pipe_chunk<BK, LA, LB>(f0, f1, s1, s1 + OPS, 1, true, wm, wn, i, h, acc);
      if (nxt) { LdA::store(s0, tid, r.ra0); LdB::store(s0 + OPS, tid, r.rb0); }
      PS_FENCE();
      pipe_chunk<BK, LA, LB>(f1, f0, s1, s1 + OPS, 2, true, wm, wn, i, h, acc);
      pipe_chunk<BK, LA, LB>(f0, f1, s1, s1 + OPS, 3, true, wm, wn, i, h, acc);
      __syncthreads();
      PS_FENCE();
      pipe_chunk<BK, LA, LB>(f1, f0, s0, s0 + OPS, 0, nxt, wm, wn, i, h, acc);
    }
  }
  // every LDS read was waited for by the barrier of the last K-tile: smem is free
}

// The same schedule with segmented accumulation, for products whose K extent is a multiple of
// SEG_K (the Newton products: npad is a multiple of 128).  The loop runs over SEGMENTS of four
// K-tiles with the flush `total += acc; acc = 0` at the end of each -- unconditional, so the two
// accumulator sets have one definition each (a flush under `if` makes the register allocator copy
// both sets at the merge point and spill ~400 registers).  ONE register set of global loads: the
// loads of K-tile t + 1 are issued right after the set has been written to LDS in the middle of
// K-tile t - 1, i.e. one K-tile of MFMA time ahead of their use (two sets: 1.25).  Writes acc
// (does not accumulate into it).
template <int BK, int LA, int LB>
__device__ inline void deep_run_pipe_seg(const Operand& A, const Operand& B, int Kext, float* smem,
                                         f32x16 (&acc)[2][2], unsigned long long* t_fill = nullptr) {
  static_assert(BK == 32 && SEG_K == 4 * BK, "a segment is four K-tiles");
  using LdA = TileLoader<LA, BK, false>;
  using LdB = TileLoader<LB, BK, false>;
  constexpr int OPS = SmemCfg<BK>::op_size(LA);
  constexpr int STG = SmemCfg<BK>::op_size(LA) + SmemCfg<BK>::op_size(LB);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 31, h = lane >> 5;
  // whole segments: the operands are zero beyond Kext up to the next multiple of 128 (the Newton
  // workspace is padded to 128), and x + 0 * 0 = x exactly
  const int nk = ((Kext + SEG_K - 1) / SEG_K) * (SEG_K / BK);
  float* s0 = smem;
  float* s1 = smem + STG;
  f32x4 ra[LdA::NV], rb[LdB::NV];
  f32x16 total[2][2];
  zero_acc(total);
  LdA::load(A, 0, tid, ra);
  LdB::load(B, 0, tid, rb);
  LdA::store(s0, tid, ra);
  LdB::store(s0 + OPS, tid, rb);
  LdA::load(A, BK, tid, ra);         // K-tile 1 (nk >= 4)
  LdB::load(B, BK, tid, rb);
  __syncthreads();
  if (t_fill != nullptr && tid == 0) *t_fill = __builtin_amdgcn_s_memrealtime();  // dev trace
  FragKM f0, f1;
  pipe_read_a<BK, LA>(s0, 0, wm, i, h, f0);
  if (LB == MC) {
#pragma unroll
    for (int s = 0; s < 4; ++s) pipe_read_b<BK>(s0 + OPS, 0, s, wn, i, h, f0);
  } else {
    pipe_read_b_kc<BK>(s0 + OPS, 0, wn, i, h, f0);
  }
  for (int kt = 0; kt < nk; kt += 4) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int k0 = kt + 2 * half;
      {  // even K-tile k0: image in s0; the set holds K-tile k0 + 1 (-> s1), then requests k0 + 2
        if (half == 0)   // first MFMAs of the segment: C = 0
          pipe_chunk<BK, LA, LB, true>(f0, f1, s0, s0 + OPS, 1, true, wm, wn, i, h, acc);
        else
          pipe_chunk<BK, LA, LB>(f0, f1, s0, s0 + OPS, 1, true, wm, wn, i, h, acc);
        LdA::store(s1, tid, ra); LdB::store(s1 + OPS, tid, rb);
        PS_FENCE();
        LdA::load(A, min(k0 + 2, nk - 1) * BK, tid, ra);   // unconditional (see deep_run)
        LdB::load(B, min(k0 + 2, nk - 1) * BK, tid, rb);
        PS_FENCE();
        pipe_chunk<BK, LA, LB>(f1, f0, s0, s0 + OPS, 2, true, wm, wn, i, h, acc);
        pipe_chunk<BK, LA, LB>(f0, f1, s0, s0 + OPS, 3, true, wm, wn, i, h, acc);
        __syncthreads();
        PS_FENCE();
        pipe_chunk<BK, LA, LB>(f1, f0, s1, s1 + OPS, 0, true, wm, wn, i, h, acc);
      }
      {  // odd K-tile k0 + 1: image in s1; the set holds K-tile k0 + 2 (-> s0), then requests k0 + 3
        const bool nxt = k0 + 2 < nk;
        pipe_chunk<BK, LA, LB>(f0, f1, s1, s1 + OPS, 1, true, wm, wn, i, h, acc);
        if (nxt) { LdA::store(s0, tid, ra); LdB::store(s0 + OPS, tid, rb); }
        PS_FENCE();
        LdA::load(A, min(k0 + 3, nk - 1) * BK, tid, ra);
        LdB::load(B, min(k0 + 3, nk - 1) * BK, tid, rb);
        PS_FENCE();
        pipe_chunk<BK, LA, LB>(f1, f0, s1, s1 + OPS, 2, true, wm, wn, i, h, acc);
        pipe_chunk<BK, LA, LB>(f0, f1, s1, s1 + OPS, 3, true, wm, wn, i, h, acc);
        __syncthreads();
        PS_FENCE();
        pipe_chunk<BK, LA, LB>(f1, f0, s0, s0 + OPS, 0, nxt, wm, wn, i, h, acc);
      }
    }
    // every MFMA of this segment is issued (the last chunk only READS the next K-tile); the next
    // segment's first MFMAs overwrite acc
    seg_add(acc, total);
    PS_FENCE();
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = total[tm][tn];
}

// Accumulating form: acc += A * B over k in [0, Kext).
// DEEP = false: the global loads of K-tile t+1 are issued at the top of iteration t and
//   written to the other LDS buffer at its bottom (one K-tile of MFMAs covers their latency).
// DEEP = true: two register sets; the loads of K-tile t+2 are issued at the top of iteration
//   t and the set holding tile t+1 (issued one iteration earlier) is written at its bottom, so
//   a load has two K-tiles of MFMA time to land.  With BK = 16 one K-tile is only 2048
//   MFMA-cycles per wavefront, less than the loaded-HBM latency: without the second set a
//   workgroup's K loop cannot go faster than one memory round trip per K-tile even when the
//   MFMA pipe is free (measured: 6900 cycles per K-tile whether 2 or 3 workgroups shared
//   the CU).  Same summation order, bit-identical results.
template <int LA, int LB, int BK, bool GUARD, bool DEEP = false, bool PIPE = false, bool SEGV = false>
__device__ inline void gemm_tile_accum(const Operand& A, const Operand& B, int Kext,
                                       float* smem, f32x16 (&acc)[2][2],
                                       unsigned long long* t_fill = nullptr, bool seg = false) {
  using LdA = TileLoader<LA, BK, GUARD>;
  using LdB = TileLoader<LB, BK, GUARD>;
  constexpr int OPS = SmemCfg<BK>::op_size(LA);
  constexpr int STG = SmemCfg<BK>::op_size(LA) + SmemCfg<BK>::op_size(LB);
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int wm = wave >> 1;
  const int wn = wave & 1;

  const int nk = (Kext + BK - 1) / BK;
  if (!DEEP) {
    f32x4 ra[LdA::NV], rb[LdB::NV];
    LdA::load(A, 0, tid, ra);
    LdB::load(B, 0, tid, rb);
    LdA::store(smem, tid, ra);
    LdB::store(smem + OPS, tid, rb);
    __syncthreads();
    constexpr int SEGT = SEG_K / BK;
    f32x16 total[SEGV ? 2 : 1][SEGV ? 2 : 1];
    if constexpr (SEGV) zero_acc(total);
    (void)total; (void)SEGT;
    for (int kt = 0; kt < nk; ++kt) {
      float* cur = smem + (kt & 1) * STG;
      float* nxt = smem + ((kt + 1) & 1) * STG;
      const bool more = kt + 1 < nk;
      if (more) {
        LdA::load(A, (kt + 1) * BK, tid, ra);
        LdB::load(B, (kt + 1) * BK, tid, rb);
      }
      compute_ktile<LA, LB, BK>(cur, cur + OPS, acc, wm, wn, lane);
      if (more) {
        LdA::store(nxt, tid, ra);
        LdB::store(nxt + OPS, tid, rb);
      }
      __syncthreads();
      if constexpr (SEGV) { if (seg && (kt + 1) % SEGT == 0 && more) seg_flush(acc, total); }
    }
    if constexpr (SEGV) { if (seg) seg_finish(acc, total); }
    return;
  }
  // ---- two register sets, loop unrolled by two so that the set indices are static ----
  DeepSets<LA, LB, BK, GUARD> sets;
  deep_issue_first<LA, LB, BK, GUARD>(A, B, Kext, sets);
  if constexpr (PIPE && BK == 32 && !GUARD)
    deep_run_pipe<BK, LA, LB>(A, B, Kext, smem, acc, sets, t_fill);
  else
    deep_run<LA, LB, BK, GUARD, SEGV>(A, B, Kext, smem, acc, sets, t_fill, seg);
}

template <int LA, int LB, int BK, bool GUARD, bool DEEP = false, bool PIPE = false, bool SEGV = false>
__device__ inline void gemm_tile(const Operand& A, const Operand& B, int Kext,
                                 float* smem, f32x16 (&acc)[2][2],
                                 unsigned long long* t_fill = nullptr, bool seg = false) {
  if constexpr (SEGV && PIPE && DEEP && BK == 32 && !GUARD) {
    // two complete loops behind one uniform branch (the caller guarantees Kext % SEG_K == 0)
    if (seg) {
      deep_run_pipe_seg<BK, LA, LB>(A, B, Kext, smem, acc, t_fill);
    } else {
      zero_acc(acc);
      gemm_tile_accum<LA, LB, BK, GUARD, DEEP, PIPE, false>(A, B, Kext, smem, acc, t_fill, false);
    }
  } else {
    zero_acc(acc);
    gemm_tile_accum<LA, LB, BK, GUARD, DEEP, PIPE, SEGV>(A, B, Kext, smem, acc, t_fill, seg);
  }
}

// Accumulator element -> (row, col) inside the 128x128 tile (C/D layout of the
// 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
__device__ inline int acc_row(int wm, int tm, int r, int lane) {
  return wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
__device__ inline int acc_col(int wn, int tn, int lane) {
  return wn * 64 + tn * 32 + (lane & 31);
}

// Mirror store for symmetric results: writes the TRANSPOSE of this workgroup's
// 128x128 accumulator tile to dst (tile origin (row0, col0) of the transposed
// position), and optionally alpha * value to dst2.  The tile is staged through LDS
// in two 64-row halves (stride 129: conflict-free both ways) so that the global
// stores are contiguous 256-byte runs.  smem must hold 64*129 floats and be free
// (gemm_tile ends with a barrier).  All 256 threads must call.
template <bool WT = false>
__device__ inline void store_tile_transposed(const f32x16 (&acc)[2][2], float* smem,
                                             float* dst, float* dst2, float alpha, int ld,
                                             int row0, int col0) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int TLD = 129;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (wm == h) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int lr = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);  // row in half
            const int c = acc_col(wn, tn, lane);
            smem[lr * TLD + c] = acc[tm][tn][r];
          }
    }
    __syncthreads();
    // transposed read: thread -> (output row = c, 64 consecutive output cols = lr)
    for (int e = tid; e < 128 * 64; e += NTHREADS) {
      const int c = e >> 6, lr = e & 63;
      const float v = smem[lr * TLD + c];
      const int64_t o = (int64_t)(row0 + c) * ld + col0 + h * 64 + lr;
      tstore1<WT>(dst + o, v);
      if (dst2 != nullptr) tstore1<WT>(dst2 + o, __fmul_rn(alpha, v));
    }
    __syncthreads();
  }
}

// Vectorised mirror store (16-byte LDS and global accesses).  Same contract as
// store_tile_transposed; smem must hold 64*132 floats.  The accumulator layout gives every
// lane 4 CONSECUTIVE tile rows per register quad at one tile column, i.e. 4 consecutive
// elements of a row of the TRANSPOSED tile: the transposed image T[c][r] is written with
// ds_write_b128 (row stride 132 floats: 8 lanes x 16 B cover the 32 banks), read back with
// ds_read_b128 along r, and stored as 512-byte runs of the mirrored tile's rows.  The tile is
// staged in two halves by accumulator column block tn (64 tile columns each, all four
// wavefronts take part in both).
template <bool WT = false>
__device__ inline void store_tile_transposed_v4(const f32x16 (&acc)[2][2], float* smem,
                                                float* dst, float* dst2, float alpha, int ld,
                                                int row0, int col0) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int TLD = 132;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    // T row ci = wn*32 + (lane&31)  <->  tile column wn*64 + h*32 + (lane&31)
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
    // thread -> (T row ci = (tid>>5) + 8k, 4 consecutive r = 4*(tid&31)..)
    const int r4 = (tid & 31) * 4;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ci = (tid >> 5) + 8 * k;
      const f32x4 v = *reinterpret_cast<const f32x4*>(smem + ci * TLD + r4);
      const int c = (ci >> 5) * 64 + h * 32 + (ci & 31);  // tile column = mirrored tile's row
      const int64_t o = (int64_t)(row0 + c) * ld + col0 + r4;
      if (WT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gstore1_wt(dst + o + e, v[e]);
      } else {
        *(f32x4 PS_GLOBAL*)(dst + o) = v;
      }
      if (dst2 != nullptr) {
        f32x4 w = {__fmul_rn(alpha, v[0]), __fmul_rn(alpha, v[1]), __fmul_rn(alpha, v[2]),
                   __fmul_rn(alpha, v[3])};
        if (WT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) gstore1_wt(dst2 + o + e, w[e]);
        } else {
          *(f32x4 PS_GLOBAL*)(dst2 + o) = w;
        }
      }
    }
    __syncthreads();
  }
}

// Bijective XCD-aware remap: consecutive logical tiles share an XCD (workgroups
// are dealt round-robin over the 8 XCDs), so the tiles of one matrix block reuse
// its operand panels out of one L2.  Speed only; any placement is correct.
__device__ inline int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// NaN-propagating max of |x| on bit patterns (jnp.max returns NaN if any NaN).
__device__ inline unsigned abs_bits(float x) {
  return __float_as_uint(x) & 0x7fffffffu;
}

__device__ inline unsigned wave_max_u32(unsigned v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

__device__ inline float wave_sum_f32(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

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

}  // namespace psk
