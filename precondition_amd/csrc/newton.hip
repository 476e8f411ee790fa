// newton.hip — batched matrix inverse p-th root by the coupled Newton iteration
// (reference: matrix_inverse_pth_root, DS:702-940, under vmap DS:2742-2744).
//
// Data layout in HBM (caller-provided workspace): every block b of effective
// size n (= min(n, padding_start)) owns ten square float32 buffers of side
// npad = round_up(n, 128), zero outside [0,n)x[0,n): M[2], H[2] (ping-pong, H[old]
// is the reference's old_mat_h), Mi = (1-alpha) I + alpha M, and five powering
// temporaries.  Zero padding makes every product tile-exact without bounds
// checks and is the reference's own padding semantics (identity and matrix are
// masked at DS:777-783).
//
// One Newton step of a block is a small dependency graph of n^3 products
// (DS:844-846; mat_power DS:655-678 with the same multiplication order, minus the
// exact "@ I" and the unused trailing square):
//     P0 : H' = H Mi                                  (needs Mi of this step)
//     P1..P(L-1) : binary-powering chain Mi^p          (each needs the previous one)
//     PL : M' = Mi^p M; epilogue writes the next Mi and max|M' - I|   (needs P(L-1), P0)
// followed by the loop control of DS:836-848 / DS:858-885 (ratio guard, retry with
// ridge*10^i).  Blocks are independent of each other.
//
// Execution A (PS_NEWTON_PERSISTENT=1) = ONE persistent kernel for the whole call
// (newton_persistent_kernel):
// a resident grid of workgroups pulls (block, product, 128x128 tile) items from
// per-XCD device queues.  The last tile of a product to arrive (an agent-scope counter)
// releases the products that depend on it; the last tile of a step evaluates the loop
// control for its block and releases the next step, a retry (re-initialisation with a
// larger ridge) or the copy-out.  There is no global barrier and no host round trip:
// a block advances as fast as its own tiles complete, blocks that converge early simply
// stop producing items, and the tail of one product overlaps the head of others.
// Inter-workgroup hand-off follows cdna_hip_programming.md Guideline 16: payload tiles
// are stored write-through (sc1), every storing wave drains (s_waitcnt vmcnt(0)), the
// workgroup's barrier, one lane's agent-scope release + counter add; the consumer
// pops an item (relaxed agent atomics), ONE agent-scope acquire, then plain loads.
// Mutable per-block loop state is only touched through agent-scope atomics.
//
// Execution B (default, the faster one today — see persistent_mode() below) = staged: one
// launch per product stage for the whole batch + a control launch per step, the host
// polling "blocks still running" one iteration behind the GPU.  Both run the same tile
// code and produce bit-identical results (tests/test_gpu_round2.py).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "gemm_core.hip.h"
#include "gemm_bf16x.hip.h"
#include "options.h"
#include "power_iter.hip.h"

namespace psk {

constexpr int NBK = 16;        // K-tile depth of the Newton products
constexpr int MAX_PROD = 8;    // products per Newton step (p <= 64)
constexpr int NTEMP = 5;
constexpr int NQ = 8;          // item queues (one per XCD)


enum Phase { PH_INIT = 0, PH_ACTIVE = 1, PH_DONE = 2 };
enum BufId { ID_MI = 0, ID_MCUR, ID_MNEXT, ID_HCUR, ID_HNEXT, ID_T0 };

// Immutable after upload.
struct NewtonBlock {
  const float* a;
  float* out;
  float* M[2];
  float* H[2];
  float* Mi;
  float* T[NTEMP];
  float* sumsq_partial;  // [npad/128 * npad/128] partial sums of ||D||_F^2
  const int* asym;       // *asym != 0: not exactly symmetric -> full products
  int n;        // effective size
  int n_full;   // rows/cols of a and out
  int lda, ldo, npad, p;
  float alpha, one_minus_alpha, inv_p;
  int nprod;    // products per step (L + 1)
  int queue;    // item queue of this block (persistent execution)
  short pa[MAX_PROD], pb[MAX_PROD], pc[MAX_PROD];  // operand / result buffer ids
};

// Mutable loop state (DS:836-848 carry + DS:858-885 retry state).  Staged execution:
// plain accesses ordered by kernel boundaries.  Persistent execution: agent-scope
// atomics only.
struct NewtonState {
  int phase, cur, it, tries, total_iters, result_sel;
  float err, ratio, max_ev, ridge, ridge_try;
  unsigned err_bits;
  // averaged M update: max |X - X^T| and max |X| of the step (bit patterns), and whether the
  // NEXT step still averages (newton_avg_next: a fixed number of leading steps)
  unsigned asym_bits, xmax_bits;
  int avg_on;
  int general;        // != 0: the block is not exactly symmetric (copy of *NewtonBlock::asym, written
                      // by newton_setup_kernel: one dependent load less at the head of every tile)
  float asym_first;   // number of steps (all tries) whose M update was averaged: PS_M_AVG_STEPS
  int power_iters;
  // per-block policy, set by the host from the call's options and the caller's iteration-count
  // hint (ps_options.iters_hint): averaged leading steps of a try, and whether the M-side products
  // are summed in segments (gemm_core.hip.h SEG_K)
  int navg, seg;
  // arrival counters of the persistent execution
  unsigned c_init1, c_init2, c_join, c_copy;
  unsigned c_prod[MAX_PROD];
};

struct NewtonTask {
  int block;
  int prod;
};

// Stage tile lists: task = block | product << 24 (no second table between a tile and its block);
// init tile list: task = block.
struct TileEntry {
  int task;
  short tm, tn;
};
constexpr int TE_BLOCK_MASK = 0xffffff;

struct HostStatus {
  int gen;
  int not_done;
  int need_init;
  int pad;
};

// Mi = (1-alpha) I + alpha M.  For p = 1 the M update multiplies by Mi itself (the
// powering chain is empty) while its epilogue writes the next Mi: those blocks alternate
// Mi between nb->Mi and the (otherwise unused) last powering temporary by step parity.
__device__ inline float* mi_buf(const NewtonBlock* nb, int cur) {
  return (nb->p == 1 && cur) ? nb->T[NTEMP - 1] : nb->Mi;
}

// z^(1/p) of DS:873.  One out-of-line copy, so that every kernel that initialises a try
// (staged and persistent execution) rounds it identically.
__device__ __attribute__((noinline)) float pth_root_of_scale(float z, float inv_p) {
  return powf(z, inv_p);
}

__device__ inline float* resolve(const NewtonBlock* nb, int id, int cur) {
  switch (id) {
    case ID_MI: return mi_buf(nb, cur);
    case ID_MCUR: return nb->M[cur];
    case ID_MNEXT: return nb->M[cur ^ 1];
    case ID_HCUR: return nb->H[cur];
    case ID_HNEXT: return nb->H[cur ^ 1];
    default: return nb->T[id - ID_T0];
  }
}

// ---- one 128x128 tile of one product --------------------------------------------
// WT: write-through stores (see PERSIST_WT).  flags:
//   TF_MIRROR   also store the transposed tile (symmetric products: only tiles tm <= tn run)
//   TF_RAW      store the product tile to C only (no M-update epilogue): first pass of an
//               averaged M update
//   TF_AVG      before the epilogue, replace the accumulators X by (X + Y^T) / 2 where Y is
//               the tile at the transposed position (tn, tm) of C, written by a TF_RAW pass
//   TF_SELFAVG  diagonal tile of an averaged M update: X is stored raw, then averaged with
//               its own transpose
// All 256 threads; ends without a barrier.
//
// Why an averaged M update (M' = (X + X^T)/2, X = Mi^p M computed in full): the iterates
// commute only up to rounding, so X carries an antisymmetric part K ~ [Mi^p, M]/2.  Copying
// the upper triangle over the lower (TF_MIRROR) turns K into a SYMMETRIC perturbation of the
// same size, which moves eigenvalues at first order, while K itself is harmless (x^T K x = 0)
// and averaging removes it.  Measured at cond 7e3, p = 4, n = 1000 (error of the root against
// float64): reference arithmetic (NumPy full products) 1.26e-4; this kernel with full products
// (PS_SYMMETRY_GENERAL) 1.96e-4, mirrored everywhere 9.3e-4, averaged M update in the first 2
// steps 2.8e-4, in the first 4 steps 1.76e-4 (the default since round 3: inside the spread of
// the two full-product float32 evaluations; +3.9 % time on 256 x 512^2 and 64 x 1024^2 over 2
// steps), first 6 steps 1.66e-4, every step 1.65e-4 (+10...14 %); on well-conditioned blocks
// all variants are at 1e-6.  The H update and the squares (exactly symmetric for
// symmetric input) stay mirrored (tools/dev_sym_accuracy.py: averaging them changes nothing).
//   TF_SELFSYM  diagonal tile of a product of two different symmetric matrices: X is replaced by
//               (X + X^T) / 2 in registers (through LDS) before it is stored
enum TileFlags { TF_MIRROR = 1, TF_RAW = 2, TF_AVG = 4, TF_SELFAVG = 8, TF_SELFSYM = 16 };

// acc <- (acc + acc^T) / 2 for a DIAGONAL 128x128 tile.  A product A B of two different
// symmetric matrices is symmetric only up to rounding inside its diagonal blocks (off the
// diagonal the mirror store makes the iterate bitwise symmetric; squares A A are bitwise
// symmetric by themselves).  With this the iterates are bitwise symmetric everywhere, which is
// what lets newton_product_tile read element (k, n) of the right operand as B[n][k].  The tile
// goes through LDS in two halves of 64 rows (image row stride 132: the layout and the read
// pattern of average_with_transposed_tile); smem: 64*132 floats, free; ends with a barrier.
__device__ inline void symmetrize_diag_tile(f32x16 (&acc)[2][2], float* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int TLD = 132;
  const f32x16 keep = acc[1][0];   // rows of half 1 / columns of half 0: changed in half 0
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    // image of the tile rows with row bit 5 == h: S[ci][col], ci = (row >> 6) * 32 + (row & 31)
    float* wrow = smem + (wm * 32 + 4 * (lane >> 5)) * TLD + wn * 64 + (lane & 31);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = (h == 1 && tn == 0) ? keep[4 * g + j] : acc[h][tn][4 * g + j];
          wrow[(8 * g + j) * TLD + tn * 32] = v;
        }
    __syncthreads();
    const float* trow = smem + (wn * 32 + (lane & 31)) * TLD + wm * 64 + 4 * (lane >> 5);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(trow + tm * 32 + 8 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[tm][h][4 * g + j] = __fmul_rn(0.5f, __fadd_rn(acc[tm][h][4 * g + j], pv[j]));
      }
    __syncthreads();
  }
}

// acc <- (acc + P^T) / 2 with P the 128x128 tile of C at (prow0, pcol0) (the transposed
// position of the accumulator tile).  P's rows are staged through LDS in two halves with the
// layout of store_tile_transposed_v4's image (T row = accumulator column), so that global
// reads are 512-byte runs and every LDS access is 16 bytes.  smem: 64*132 floats.
__device__ inline void average_with_transposed_tile(f32x16 (&acc)[2][2], float* smem,
                                                     const float* C, int ld, int prow0,
                                                     int pcol0, NewtonState* st) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  constexpr int TLD = 132;
  const int r4 = (tid & 31) * 4;
  unsigned dmax = 0, xmax = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ci = (tid >> 5) + 8 * k;
      const int c = (ci >> 5) * 64 + h * 32 + (ci & 31);
      *reinterpret_cast<f32x4*>(smem + ci * TLD + r4) =
          gload4(C + (int64_t)(prow0 + c) * ld + pcol0 + r4);
    }
    __syncthreads();
    const float* trow = smem + (wn * 32 + (lane & 31)) * TLD + wm * 64 + 4 * (lane >> 5);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(trow + tm * 32 + 8 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x = acc[tm][h][4 * g + j];
          const unsigned d = abs_bits(__fsub_rn(x, pv[j])), m = abs_bits(x);
          dmax = d > dmax ? d : dmax;
          xmax = m > xmax ? m : xmax;
          acc[tm][h][4 * g + j] = __fmul_rn(0.5f, __fadd_rn(x, pv[j]));
        }
      }
    __syncthreads();
  }
  // how far the computed product is from symmetric (the size of the commutator noise the
  // averaging removes): steers whether the next steps still need the full product
  dmax = wave_max_u32(dmax);
  xmax = wave_max_u32(xmax);
  if (lane == 0) { atomicMax(&st->asym_bits, dmax); atomicMax(&st->xmax_bits, xmax); }
}

template <bool WT>
__device__ __forceinline__ void newton_tile_epilogue(const NewtonBlock* nb, NewtonState* st, int prod,
                                                     int cur, int tm_, int tn_, int flags,
                                                     float* smem, f32x16 (&acc)[2][2]);

template <int BK, bool WT, bool DEEP, int XM = 0, bool PIPE = false, bool CAREFUL = false>
__device__ __forceinline__ void newton_product_tile(const NewtonBlock* nb, NewtonState* st, int prod,
                                           int cur, int tm_, int tn_, int flags,
                                           float* smem, unsigned long long* stamp = nullptr) {
  const int n = nb->n, ld = nb->npad;
  Operand A{resolve(nb, nb->pa[prod], cur), ld, tm_ * TILE, ld, ld, true};
  Operand B{resolve(nb, nb->pb[prod], cur), ld, tn_ * TILE, ld, ld, true};
  f32x16 acc[2][2];
  // Near convergence (max|M - I| < 1e-3 at the start of the step) the split products' noise floor
  // (~2e-6 in max|M - I| on cond 1e4 blocks) is above the 1e-6 stop threshold of DS:836, so the
  // last steps of a block run the exact float32 products: same stop decisions as the parity path.
  // Segmented accumulation (ps_options.accumulation): every product of a block that the caller's
  // hint does not mark well conditioned (the M side decides the error of p = 4 roots, the H update
  // that of p = 2 roots: tools/dev_chain_accuracy.py, dev_r4_newton.py).  Only the CAREFUL
  // instantiations carry the second accumulator set (the driver launches them when the call has
  // at least one such block); a block's arithmetic does not depend on what else is in the call.
  const bool seg = CAREFUL && st->seg != 0;
  if (XM == 6 && st->general == 0 && st->err > 1e-3f)
    gemm_tile_bf16x_sym<6>(A.p, ld, A.mn0, B.p, ld, B.mn0, ld, smem, acc);
  else if (XM == 3 && st->general == 0 && st->err > 3e-2f)
    gemm_tile_bf16x_sym<3>(A.p, ld, A.mn0, B.p, ld, B.mn0, ld, smem, acc);
  else if (st->general == 0)
    // Exactly symmetric input block (sym_check): element (k, n) of the right operand is read as
    // B[n][k], i.e. both operands are staged k-contiguous and every fragment read is a 16-byte
    // ds_read with an immediate offset -- 4 LDS instructions per 16 MFMAs instead of 6-10 and no
    // address arithmetic in the K loop (VALU work costs fp32-MFMA cycles: tools/dev_mfma_mix.py).
    // Exact because the iterates are bitwise symmetric: mirror store off the diagonal,
    // symmetrize_diag_tile inside the diagonal tiles (section 4a).  Same k order as the
    // mn-contiguous path; every execution (staged, persistent, all K-loop variants) takes it.
    gemm_tile<KC, KC, BK, false, DEEP, PIPE, CAREFUL>(A, B, n, smem, acc, stamp ? stamp + 1 : nullptr, seg);
  else
    gemm_tile<KC, MC, BK, false, DEEP, PIPE, CAREFUL>(A, B, n, smem, acc, stamp ? stamp + 1 : nullptr, seg);

  if (stamp != nullptr && threadIdx.x == 0) *stamp = __builtin_amdgcn_s_memrealtime();  // dev profile
  newton_tile_epilogue<WT>(nb, st, prod, cur, tm_, tn_, flags, smem, acc);
}

// Epilogue of one product tile: stores (direct + mirrored), the averaged M update, and for the
// M update the next Mi and the error.  All 256 threads; smem free; ends without a barrier.
template <bool WT>
__device__ __forceinline__ void newton_tile_epilogue(const NewtonBlock* nb, NewtonState* st, int prod,
                                                     int cur, int tm_, int tn_, int flags,
                                                     float* smem, f32x16 (&acc)[2][2]) {
  const bool mirror = (flags & TF_MIRROR) != 0;
  const int n = nb->n, ld = nb->npad;
  float* C = resolve(nb, nb->pc[prod], cur);
  // Addressing of the direct stores: everything but a per-lane 32-bit offset is wave-uniform
  // (kept in SGPRs), so the 64 (or 128) stores of the epilogue need no per-store address
  // registers: element r of block (tm, tn) of this wavefront lies at
  //   C[(tile_row0 + wm*64 + tm*32 + (r&3) + 8*(r>>2)) * ld + tile_col0 + wn*64 + tn*32] + lane_off
  // with lane_off = 4*(lane>>5)*ld + (lane&31).
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lane_off = 4 * (lane >> 5) * ld + (lane & 31);
  const int lane_row = 4 * (lane >> 5), lane_col = lane & 31;
  if ((flags & TF_SELFSYM) != 0) symmetrize_diag_tile(acc, smem);
  const bool plain = prod != nb->nprod - 1 || (flags & TF_RAW) != 0;
  if (plain || (flags & TF_SELFAVG) != 0) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        float* blk = C + (int64_t)(tm_ * TILE + wm * 64 + tm * 32) * ld + tn_ * TILE + wn * 64 + tn * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tstore1<WT>(blk + (int64_t)((r & 3) + 8 * (r >> 2)) * ld + lane_off, acc[tm][tn][r]);
      }
    if (plain) {
      if (mirror)
        store_tile_transposed_v4<WT>(acc, smem, C, nullptr, 0.f, ld, tn_ * TILE, tm_ * TILE);
      return;
    }
    // TF_SELFAVG: the raw tile is in C; every lane's stores must have landed before the
    // transposed read-back (same CU: workgroup-scope visibility through its own L1/L2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if ((flags & (TF_AVG | TF_SELFAVG)) != 0)
    average_with_transposed_tile(acc, smem, C, ld, tn_ * TILE, tm_ * TILE, st);
  // M update: C = new M; Mi = (1-alpha) I + alpha M (DS:844, two roundings as
  // written there); err = max |M - I| (DS:847) with the identity masked to n.
  float* Mi = mi_buf(nb, cur ^ 1);  // the Mi of the NEXT step
  const float alpha = nb->alpha, oma = nb->one_minus_alpha;
  unsigned emax = 0;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int brow = tm_ * TILE + wm * 64 + tm * 32, bcol = tn_ * TILE + wn * 64 + tn * 32;
      float* blk = C + (int64_t)brow * ld + bcol;
      float* blk_i = Mi + (int64_t)brow * ld + bcol;
      // the identity touches this block only if it straddles the diagonal (uniform test)
      const bool on_diag = brow == bcol;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2);
        const float v = acc[tm][tn][r];
        const float ident =
            (on_diag && rr + lane_row == lane_col && brow + rr + lane_row < n) ? 1.f : 0.f;
        tstore1<WT>(blk + (int64_t)rr * ld + lane_off, v);
        tstore1<WT>(blk_i + (int64_t)rr * ld + lane_off,
                    __fadd_rn(__fmul_rn(oma, ident), __fmul_rn(alpha, v)));
        const unsigned e = abs_bits(__fsub_rn(v, ident));
        emax = e > emax ? e : emax;
      }
    }
  emax = wave_max_u32(emax);
  unsigned* red = reinterpret_cast<unsigned*>(smem);
  if (lane == 0) red[wave] = emax;
  __syncthreads();
  if (tid == 0) {
    unsigned m = red[0];
    m = red[1] > m ? red[1] : m;
    m = red[2] > m ? red[2] : m;
    m = red[3] > m ? red[3] : m;
    atomicMax(&st->err_bits, m);
  }
  if (mirror) {
    __syncthreads();  // red[] lives in smem
    // off-diagonal tile: identity is 0 there, so Mi = fl(alpha * M)
    store_tile_transposed_v4<WT>(acc, smem, C, Mi, alpha, ld, tn_ * TILE, tm_ * TILE);
  }
}

// Tile (tm <= tn) of a product, both executions.  Blocks that are not exactly symmetric
// compute the transposed position as a product of its own; symmetric blocks mirror it, except
// the M update of the first steps of a try (avg), which is computed in full and averaged
// with its transpose (see TileFlags).
template <int BK, bool WT, bool DEEP, int XM = 0, bool PIPE = false, bool CAREFUL = false>
__device__ __forceinline__ void newton_product_item(const NewtonBlock* nb, NewtonState* st, int prod,
                                           int cur, int avg, int tm, int tn, float* smem,
                                           unsigned long long* stamp = nullptr) {
  const bool sym = st->general == 0;
  int passes = 1, f0 = 0, f1 = 0;
  if (!sym) {
    passes = tm != tn ? 2 : 1;
  } else if (!(avg && prod == nb->nprod - 1)) {
    f0 = tm != tn ? TF_MIRROR : (nb->pa[prod] != nb->pb[prod] ? TF_SELFSYM : 0);
  } else if (tm == tn) {
    f0 = TF_SELFAVG;
  } else {
    passes = 2; f0 = TF_RAW; f1 = TF_AVG | TF_MIRROR;
  }
  // one copy of the tile code (a rolled loop): two inlined copies cost 90 VGPRs
#pragma unroll 1
  for (int pass = 0; pass < passes; ++pass) {
    if (pass) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first pass's tile is re-read
      __syncthreads();
    }
    newton_product_tile<BK, WT, DEEP, XM, PIPE, CAREFUL>(nb, st, prod, cur, pass ? tn : tm, pass ? tm : tn,
                                          pass ? f1 : f0, smem, stamp);
  }
}

// Does step `it` (the one about to run) average its M update?  The first `navg` (4) steps of a
// try do, and -- only when the opt-in threshold `thr` (PS_NEWTON_AVG_ERR, default 0 = off) is
// set -- steps after the second stop early once the error max|M - I| has dropped below it.
// Measured (tools/dev_avg_sweep.py, dev_avg_adaptive.py): root error vs float64 of the cond 7e3,
// p = 4, 1024^2 block of the ViT-B tree 2.8e-4 with 2 averaged steps, 1.76e-4 with 4, 1.66e-4
// with 6 (the oracle's full products 1.26e-4); a cond-10 Wishart block is 1e-6 from float64 with
// any count.  The elementwise error does NOT separate those two block classes early enough to
// steer the count (0.69 vs 0.81 after two steps, both falling), which is why the default stays
// the fixed 4 (about +4 % on the headline) and the threshold is an opt-in.  The number of
// averaged steps is reported per block in the metrics table (PS_M_AVG_STEPS) for FLOP accounting.
__device__ inline bool newton_avg_next(int it, int navg, float err, float thr) {
  return it < 2 ? it < navg : (it < navg && err > thr);
}

// ---- (re)initialisation of a try (DS:866-875), tile bodies ----------------------------
// pass 1: partial sum of squares of D = A + ridge_try * I over one 128x128 tile.
template <bool WT>
__device__ inline void newton_init1_tile(const NewtonBlock* nb, float ridge_try, int tm,
                                         int tn, float* red) {
  const int n = nb->n, tid = threadIdx.x;
  float ss = 0.f;
  const int tpr = nb->npad / TILE;
  // a thread's 64 elements (column tid % 128, every second row) in the same order as ever -- the sum's rounding is
  // part of z -- but eight loads in flight per trip
  const float* a = nb->a;
  const int64_t lda = nb->lda;
  const int col = tn * TILE + (tid & 127), rbase = tm * TILE + (tid >> 7);
  for (int k0 = 0; k0 < TILE / 2; k0 += 8) {
    float d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = rbase + 2 * (k0 + u);
      d[u] = (row < n && col < n) ? gload1(a + (int64_t)row * lda + col) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = rbase + 2 * (k0 + u);
      if (row < n && col < n) {
        float x = d[u];
        if (row == col) x = __fadd_rn(x, ridge_try);
        ss += x * x;
      }
    }
  }
  ss = wave_sum_f32(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  if (tid == 0)
    tstore1<WT>(nb->sumsq_partial + tm * tpr + tn, ((red[0] + red[1]) + red[2]) + red[3]);
}

// pass 2: z, M0 = D z, H0 = I z^(1/p), Mi0, err0 = max|M0 - I|.
template <bool WT>
__device__ inline void newton_init2_tile(const NewtonBlock* nb, NewtonState* st,
                                         float ridge_try, int cur, int tm, int tn,
                                         float* scratch) {
  float* bc = scratch;
  unsigned* red = reinterpret_cast<unsigned*>(scratch + 4);
  const int n = nb->n, ld = nb->npad, tid = threadIdx.x;
  const int tpr = ld / TILE;
  const float sumsq = fixed_order_sum_wave0(nb->sumsq_partial, tpr * tpr, tid, bc);
  const float z = __fdiv_rn((float)(1 + nb->p), __fmul_rn(2.f, sqrtf(sumsq)));  // DS:870
  const float h0 = pth_root_of_scale(z, nb->inv_p);                             // DS:873
  const float rt = ridge_try, alpha = nb->alpha, oma = nb->one_minus_alpha;
  float* M = nb->M[cur];
  float* H0 = nb->H[cur];
  float* H1 = nb->H[cur ^ 1];
  float* Mi = mi_buf(nb, cur);
  unsigned emax = 0;
  // (eight loads in flight per trip: behind the stores of the trip before, the compiler issues them one at a time)
  const float* a = nb->a;
  const int64_t lda = nb->lda;
  const int col = tn * TILE + (tid & 127), rbase = tm * TILE + (tid >> 7);
  for (int k0 = 0; k0 < TILE / 2; k0 += 8) {
    float dv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = rbase + 2 * (k0 + u);
      dv[u] = (row < n && col < n) ? gload1(a + (int64_t)row * lda + col) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
    const int row = rbase + 2 * (k0 + u);
    float m = 0.f, mi = 0.f, h = 0.f, h1 = 0.f;
    if (row < n && col < n) {
      float d = dv[u];
      const float ident = row == col ? 1.f : 0.f;
      if (row == col) d = __fadd_rn(d, rt);                 // DS:869
      m = __fmul_rn(d, z);                                   // DS:871
      mi = __fadd_rn(__fmul_rn(oma, ident), __fmul_rn(alpha, m));
      h = __fmul_rn(ident, h0);
      // The first H update of a try, H0 Mi with H0 = h0 I (DS:845), is one product per
      // element: fl(h0 * mi) -- what the MFMA returns for a single non-zero term -- so it is
      // written here and the step-0 tiles of product P0 are skipped (stage kernel / queue).
      h1 = __fmul_rn(h0, mi);
      const unsigned eb = abs_bits(__fsub_rn(m, ident));     // DS:872
      emax = eb > emax ? eb : emax;
    }
    const int64_t o = (int64_t)row * ld + col;
    tstore1<WT>(M + o, m);
    tstore1<WT>(Mi + o, mi);
    tstore1<WT>(H0 + o, h);
    tstore1<WT>(H1 + o, h1);
    }
  }
  emax = wave_max_u32(emax);
  if ((tid & 63) == 0) red[tid >> 6] = emax;
  __syncthreads();
  if (tid == 0) {
    unsigned m = red[0];
    m = red[1] > m ? red[1] : m;
    m = red[2] > m ? red[2] : m;
    m = red[3] > m ? red[3] : m;
    atomicMax(&st->err_bits, m);
  }
}

// Copy-out of one 128x128 tile of the result (DS:902-907): H[sel] cropped to n, zero
// beyond it up to n_full.
__device__ inline void newton_copy_tile(const NewtonBlock* nb, int sel, int tm, int tn) {
  const int n = nb->n, nf = nb->n_full, tid = threadIdx.x;
  const float* H = nb->H[sel];
  for (int e = tid; e < TILE * TILE; e += 256) {
    const int row = tm * TILE + e / TILE, col = tn * TILE + e % TILE;
    if (row < nf && col < nf)
      gstore1(nb->out + (int64_t)row * nb->ldo + col,
              (row < n && col < n) ? gload1(H + (int64_t)row * nb->npad + col) : 0.f);
  }
}

__device__ inline void write_metrics(float* metrics, int b, float err, int it, float ratio,
                                     int tries, int total_iters, float max_ev, int pit,
                                     float asym_first = 0.f) {
  float* m = metrics + (int64_t)b * PS_METRICS_STRIDE;
  m[PS_M_ERROR] = err; m[PS_M_ITERS] = (float)it; m[PS_M_ERROR_RATIO] = ratio;
  m[PS_M_RETRIES] = (float)tries; m[PS_M_TOTAL_ITERS] = (float)total_iters;
  m[PS_M_MAX_EV] = max_ev; m[PS_M_POWER_ITERS] = (float)pit; m[PS_M_AVG_STEPS] = asym_first;
}

// ==================================================================================
// staged execution (PS_NEWTON_PERSISTENT=0)
// ==================================================================================
// DEEP = false: 3 workgroups per CU.  DEEP = true: two-K-tile-deep register prefetch, 2 per CU.
// TRACE (dev, PS_NEWTON_TRACE=<file>): one record of 8 x u64 per tile -- launch sequence number,
// tile index | product << 32, HW_ID | XCC_ID << 32, and the 100 MHz clock at the start of the
// tile, after its first LDS fill, at the end of its (last) K loop and at its end.
// CAREFUL: segmented accumulation for the blocks that ask for it (gemm_core.hip.h
// deep_run_pipe_seg: one register set of loads, the other set's registers hold the segment totals;
// 251 VGPRs, no spills, two workgroups per CU like the default instantiation).
template <int BK, bool DEEP, int XM = 0, bool TRACE = false, bool PIPE = false, bool CAREFUL = false>
__global__ __launch_bounds__(256, CAREFUL ? 2 : ((DEEP && BK == 32) ? 2 : 3)) void newton_stage_kernel(
    const NewtonBlock* blocks, NewtonState* states,
    const TileEntry* tiles, int ntiles, int navg, unsigned long long* trace = nullptr,
    unsigned trace_seq = 0, unsigned trace_cap = 0) {
  extern __shared__ __align__(16) float smem[];  // SmemCfg<BK>::TOTAL floats
  __shared__ unsigned long long s_trace[2];
  // navg carries the dev stagger word here: (mode << 16) | microseconds.  Co-resident workgroups of
  // equal tiles start together and stay in lockstep (prologues, K loops and epilogues coincide);
  // delaying half of the first residents by half a tile de-phases them for the whole launch.
  if (navg != 0 && blockIdx.x < 512) {
    const int mode = navg >> 16, us = navg & 0xffff, i0 = blockIdx.x;
    // mode 2: the second half of the first residents (the dispatcher fills slot 0 of every CU with
    // workgroups 0 .. 255, then slot 1); modes 0 / 1: other halves, for A/B runs
    const bool late = mode == 2 ? i0 >= 256 : mode == 1 ? ((i0 >> 3) & 1) != 0 : (i0 & 1) != 0;
    if (late) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
    }
  }
  // gridDim.x == ntiles: one tile per workgroup (hardware dispatch).  A smaller grid
  // (PS_NEWTON_GRID) walks the list with stride gridDim.x; the next tile's descriptors are
  // loaded while the current tile computes.
  const int stride = gridDim.x;
  int i = blockIdx.x;
  TileEntry te = tiles[xcd_remap(i, ntiles)];
  NewtonTask tk{te.task & TE_BLOCK_MASK, te.task >> 24};
  while (true) {
    const int inext = i + stride;
    const bool more = inext < ntiles;
    TileEntry te_n = te;
    if (more) te_n = tiles[xcd_remap(inext, ntiles)];
    const NewtonBlock* nb = &blocks[tk.block];
    NewtonState* st = &states[tk.block];
    unsigned long long t0 = 0;
    if (TRACE) {
      t0 = __builtin_amdgcn_s_memrealtime();
      if (threadIdx.x == 0) s_trace[0] = s_trace[1] = 0;
    }
    // P0 of step 0 (H0 Mi, H0 a scaled identity) was written by newton_init2_tile
    if (st->phase == PH_ACTIVE && !(tk.prod == 0 && st->it == 0))
      newton_product_item<BK, false, DEEP, XM, PIPE, CAREFUL>(nb, st, tk.prod, st->cur, st->avg_on, te.tm,
                                                     te.tn, smem, TRACE ? s_trace : nullptr);
    if (TRACE && threadIdx.x == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile's stores have left the CU
      const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
      const unsigned rec = atomicAdd(reinterpret_cast<unsigned*>(trace), 1u);
      if (rec < trace_cap) {
        unsigned long long* r = trace + 8 + 8 * (size_t)rec;
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        r[0] = trace_seq;
        r[1] = (unsigned long long)(unsigned)i | ((unsigned long long)(unsigned)tk.prod << 32);
        r[2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
        r[3] = t0; r[4] = s_trace[1]; r[5] = s_trace[0]; r[6] = t3;
        r[7] = (unsigned long long)(unsigned)te.tm | ((unsigned long long)(unsigned)te.tn << 32);
      }
    }
    if (!more) break;
    tk = NewtonTask{te_n.task & TE_BLOCK_MASK, te_n.task >> 24};
    te = te_n;
    i = inext;
    __syncthreads();  // smem (reduction scratch of the epilogue) is reused by the next tile
  }
}

__global__ __launch_bounds__(256) void newton_init1_kernel(const NewtonBlock* blocks,
                                                           const NewtonState* states,
                                                           const TileEntry* tiles) {
  __shared__ float red[4];
  const TileEntry te = tiles[blockIdx.x];
  if (states[te.task].phase != PH_INIT) return;
  newton_init1_tile<false>(&blocks[te.task], states[te.task].ridge_try, te.tm, te.tn, red);
}

__global__ __launch_bounds__(256) void newton_init2_kernel(const NewtonBlock* blocks,
                                                           NewtonState* states,
                                                           const TileEntry* tiles) {
  __shared__ float scratch[8];
  const TileEntry te = tiles[blockIdx.x];
  NewtonState* st = &states[te.task];
  if (st->phase != PH_INIT) return;
  newton_init2_tile<false>(&blocks[te.task], st, st->ridge_try, st->cur, te.tm, te.tn, scratch);
}

// ---- loop control (single workgroup, one thread per block, grid-stride) --------
__device__ inline void finish_try(NewtonState* st) {
  // DS:878-882: error, is_converged select, retry decision.
  const bool conv = st->ratio < 1.2f;
  st->result_sel = conv ? st->cur : (st->cur ^ 1);
  st->tries += 1;
  if (st->err > 0.05f && st->tries < 6) {
    st->phase = PH_INIT;
    const float pow10[6] = {1.f, 10.f, 100.f, 1000.f, 10000.f, 100000.f};
    st->ridge_try = __fmul_rn(st->ridge, pow10[st->tries]);  // DS:869
  } else {
    st->phase = PH_DONE;
  }
}

// mode 0: after init2 (enter the inner loop, DS:874-877); mode 1: after one
// Newton step (DS:848 carry + DS:836-840 condition).
// ids (may be NULL = blocks 0 .. nblocks - 1): the blocks of one stream group (newton_driver).
__global__ __launch_bounds__(256) void newton_control_kernel(
    NewtonState* states, int nblocks, int mode, int num_iters, float tol, int gen,
    HostStatus* status, float avg_thr, const int* ids = nullptr) {
  __shared__ int s_nd, s_ni;
  if (threadIdx.x == 0) { s_nd = 0; s_ni = 0; }
  __syncthreads();
  int nd = 0, ni = 0;
  for (int k = threadIdx.x; k < nblocks; k += blockDim.x) {
    const int b = ids ? ids[k] : k;
    NewtonState* st = &states[b];
    if (mode == 0 && st->phase == PH_INIT) {
      st->err = __uint_as_float(st->err_bits);
      st->err_bits = 0;
      st->ratio = 1.f;
      st->it = 0;
      st->asym_bits = 0; st->xmax_bits = 0;
      // (blocks on the general path run full products: nothing to average, nothing counted)
      st->avg_on = (st->general == 0 && newton_avg_next(0, st->navg, st->err, avg_thr)) ? 1 : 0;
      const bool cont = st->it < num_iters && st->err > tol && st->ratio < 1.2f;
      if (cont) st->phase = PH_ACTIVE; else finish_try(st);
    } else if (mode == 1 && st->phase == PH_ACTIVE) {
      const float new_err = __uint_as_float(st->err_bits);
      st->err_bits = 0;
      st->ratio = __fdiv_rn(new_err, st->err);
      st->err = new_err;
      st->it += 1;
      st->total_iters += 1;
      st->cur ^= 1;
      if (st->avg_on) st->asym_first += 1.f;   // averaged steps so far (all tries)
      st->avg_on = (st->avg_on && newton_avg_next(st->it, st->navg, st->err, avg_thr)) ? 1 : 0;
      st->asym_bits = 0; st->xmax_bits = 0;
      const bool cont = st->it < num_iters && st->err > tol && st->ratio < 1.2f;
      if (!cont) finish_try(st);
    }
    nd += st->phase != PH_DONE;
    ni += st->phase == PH_INIT;
  }
  atomicAdd(&s_nd, nd);
  atomicAdd(&s_ni, ni);
  __syncthreads();
  if (threadIdx.x == 0 && status != nullptr) {
    status->not_done = s_nd;
    status->need_init = s_ni;
    __threadfence_system();
    status->gen = gen;
  }
}

// After the power iteration: ridge = ridge_epsilon * max(max_ev, 1e-25) (DS:830).
__global__ void newton_setup_kernel(NewtonState* states, const PiBlock* pis, int nblocks,
                                    float ridge_epsilon, int relative,
                                    const float* max_ev_given) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  NewtonState* st = &states[b];
  float max_ev = 1.f;
  int pit = 0;
  if (relative) {
    if (max_ev_given) max_ev = max_ev_given[b];  // DS:815-817: lobpcg already has it
    else { max_ev = pis[b].lambda; pit = pis[b].iters; }
  }
  st->max_ev = max_ev;
  st->general = *pis[b].asym;
  st->power_iters = pit;
  st->ridge = __fmul_rn(ridge_epsilon, fmaxf(max_ev, 1e-25f));
  // fmaxf drops a NaN max_ev; jnp.maximum propagates it.
  if (max_ev != max_ev) st->ridge = max_ev;
  st->ridge_try = st->ridge;  // * 10^0
}

// ---- result copy-out + metrics table (DS:902-907, 930-939), staged execution ------
__global__ __launch_bounds__(256) void newton_final_kernel(const NewtonBlock* blocks,
                                                           const NewtonState* states,
                                                           float* metrics) {
  const NewtonBlock* nb = &blocks[blockIdx.x];
  const NewtonState* st = &states[blockIdx.x];
  const int n = nb->n, nf = nb->n_full;
  const int tid = blockIdx.y * 256 + threadIdx.x, nth = gridDim.y * 256;
  float* out = nb->out;
  if (n == 0) {
    for (int64_t e = tid; e < (int64_t)nf * nf; e += nth) {
      const int row = e / nf, col = e % nf;
      out[(int64_t)row * nb->ldo + col] = 0.f;
    }
  } else if ((nf & 3) == 0 && (nb->ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && nf <= 16384) {
    // float4 rows (H's stride npad is a multiple of 128, its base 256-byte aligned), 32-bit index arithmetic, four
    // float4 in flight per thread: the element-wise loop below ran at 3.5 TB/s (a 64-bit division per element)
    const float* H = nb->H[st->result_sel];
    const int q = nf >> 2, total = q * nf, npad = nb->npad;
    const int64_t ldo = nb->ldo;
    for (int e0 = tid; e0 < total; e0 += 4 * nth) {
      f32x4 v[4];
      int row[4], col[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * nth;
        row[u] = e / q; col[u] = 4 * (e - row[u] * q);
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (e < total && row[u] < n) {
          if (col[u] + 3 < n) v[u] = gload4(H + (int64_t)row[u] * npad + col[u]);
          else
#pragma unroll
            for (int k = 0; k < 4; ++k) if (col[u] + k < n) v[u][k] = gload1(H + (int64_t)row[u] * npad + col[u] + k);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e0 + u * nth < total) *(f32x4 PS_GLOBAL*)(out + (int64_t)row[u] * ldo + col[u]) = v[u];
    }
  } else {
    const float* H = nb->H[st->result_sel];
    for (int64_t e = tid; e < (int64_t)nf * nf; e += nth) {
      const int row = e / nf, col = e % nf;
      out[(int64_t)row * nb->ldo + col] =
          (row < n && col < n) ? H[(int64_t)row * nb->npad + col] : 0.f;
    }
  }
  if (tid == 0) {
    if (n == 0)  // all padding: DS:930-937 (error forced to 0)
      write_metrics(metrics, blockIdx.x, 0.f, 0, 1.f, 1, 0, st->max_ev, st->power_iters);
    else
      write_metrics(metrics, blockIdx.x, st->err, st->it, st->ratio, st->tries,
                    st->total_iters, st->max_ev, st->power_iters, st->asym_first);
  }
}

// ==================================================================================
// persistent dataflow execution
// ==================================================================================
enum ItemKind { IT_INIT1 = 1, IT_INIT2 = 2, IT_PROD = 3, IT_COPY = 4, IT_EXIT = 5 };

// Queue slot = one 8-byte granule {tag, payload}, written by ONE agent-scope store:
//   [63:48] tag = (index / capacity + 1) & 0xffff     (0 = never written)
//   [47:32] block   [31:28] kind   [27:24] product   [23:16] tm   [15:8] tn   [1] averaged
//   M update in this step   [0] cur / sel
typedef unsigned long long u64;

struct PControl {
  unsigned head[NQ * 32];   // one 128-byte line per word
  unsigned tail[NQ * 32];
  unsigned blocks_done[32];
  unsigned abort_flag[32];
};

struct PArgs {
  const NewtonBlock* blocks;
  NewtonState* states;
  PControl* ctl;
  u64* slots;         // [NQ][qcap]
  float* metrics;
  unsigned qcap;      // power of two
  int nblocks;
  int nlive;          // blocks with n >= 1
  int num_iters;
  float tol;
  float avg_thr;      // ... which last while max|M - I| is above this
  int grid;           // workgroups of the persistent launch (exit tokens per queue)
  int nq;             // queues in use (1..NQ); workgroup w serves queue w % nq
  u64* prof;          // dev (PS_NEWTON_PROF=1): [grid][8] per-workgroup time split, else NULL
};

#define PS_RLX __ATOMIC_RELAXED
#define PS_AGENT __HIP_MEMORY_SCOPE_AGENT
template <typename T>
__device__ inline T ald(T* p) { return __hip_atomic_load(p, PS_RLX, PS_AGENT); }
template <typename T>
__device__ inline void ast(T* p, T v) { __hip_atomic_store(p, v, PS_RLX, PS_AGENT); }

__device__ inline u64 make_item(int block, int kind, int prod, int tm, int tn, int bit) {
  return ((u64)(unsigned)block << 32) | ((u64)kind << 28) | ((u64)prod << 24) |
         ((u64)tm << 16) | ((u64)tn << 8) | (u64)(bit & 3);
}

// Everything this lane stored before (state words, and — through the barrier that
// precedes every call site — the workgroup's drained payload tiles) becomes visible at
// agent scope before anything stored afterwards (the queue slots).
__device__ inline void release_agent() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler may drop its own wait
}

// Reserve `count` consecutive slots of queue q; one lane.
__device__ inline unsigned q_reserve(const PArgs& pa, int q, unsigned count) {
  return __hip_atomic_fetch_add(&pa.ctl->tail[q * 32], count, PS_RLX, PS_AGENT);
}
__device__ inline void q_put(const PArgs& pa, int q, unsigned pos, u64 payload) {
  const u64 tag = (u64)(((pos / pa.qcap) + 1u) & 0xffffu);
  ast(pa.slots + (size_t)q * pa.qcap + (pos & (pa.qcap - 1)), (tag << 48) | payload);
}

// Push every tile of one work unit of block b (one lane).  upper: tiles with tm <= tn only.
__device__ inline void push_tiles(const PArgs& pa, const NewtonBlock* nb, int b, int kind,
                                  int prod, int t, bool upper, int bit) {
  const unsigned count = upper ? (unsigned)(t * (t + 1) / 2) : (unsigned)(t * t);
  unsigned pos = q_reserve(pa, nb->queue, count);
  for (int tm = 0; tm < t; ++tm)
    for (int tn = upper ? tm : 0; tn < t; ++tn)
      q_put(pa, nb->queue, pos++, make_item(b, kind, prod, tm, tn, bit));
}

__device__ inline void push_product(const PArgs& pa, const NewtonBlock* nb, int b, int prod,
                                    int cur) {
  push_tiles(pa, nb, b, IT_PROD, prod, nb->npad / TILE, true, cur);
}

constexpr u64 ITEM_EXIT = ~0ull;
constexpr unsigned long long SPIN_LIMIT_TICKS = 400000000ull;  // 4 s of s_memrealtime (100 MHz)

// Dequeue = ONE returning agent-scope add on the queue's head word (a ticket), then the
// ticket's slot is polled until its tag says "published".  No compare-and-swap loop: with
// ~100 workgroups per queue finishing tiles in bursts a CAS dequeue spent 28 us per item in
// retries.  A ticket commits the workgroup to its queue; it is released at the end of the
// call by the exit tokens that the last block to finish pushes into every queue.
__device__ inline u64 q_pop_wait(const PArgs& pa, int q) {
  const unsigned t = __hip_atomic_fetch_add(&pa.ctl->head[q * 32], 1u, PS_RLX, PS_AGENT);
  u64* slot = pa.slots + (size_t)q * pa.qcap + (t & (pa.qcap - 1));
  const unsigned want = ((t / pa.qcap) + 1u) & 0xffffu;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    const u64 s = ald(slot);
    if ((unsigned)(s >> 48) == want) return s;
    if (ald(&pa.ctl->abort_flag[0]) != 0u) return ITEM_EXIT;
    if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS) {
      ast(&pa.ctl->abort_flag[0], 1u);  // give up: never hang the device
      return ITEM_EXIT;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

// ---- loop control of one block, run by the lane whose tile arrived last -----------------
__device__ inline void p_block_done(const PArgs& pa, const NewtonBlock* nb, NewtonState* st,
                                    int b, float err, int it, float ratio, int tries,
                                    int total_iters, int sel) {
  write_metrics(pa.metrics, b, err, it, ratio, tries, total_iters, ald(&st->max_ev),
                ald(&st->power_iters), ald(&st->asym_first));
  ast(&st->phase, (int)PH_DONE);
  release_agent();
  push_tiles(pa, nb, b, IT_COPY, 0, (nb->n_full + TILE - 1) / TILE, false, sel);
}

__device__ inline void p_finish_try(const PArgs& pa, const NewtonBlock* nb, NewtonState* st,
                                    int b, float err, int it, float ratio, int cur,
                                    int total_iters) {
  // DS:878-882
  const bool conv = ratio < 1.2f;
  const int sel = conv ? cur : (cur ^ 1);
  const int tries = ald(&st->tries) + 1;
  ast(&st->tries, tries);
  ast(&st->result_sel, sel);
  if (err > 0.05f && tries < 6) {
    const float pow10[6] = {1.f, 10.f, 100.f, 1000.f, 10000.f, 100000.f};
    ast(&st->ridge_try, __fmul_rn(ald(&st->ridge), pow10[tries]));  // DS:869
    ast(&st->phase, (int)PH_INIT);
    release_agent();
    push_tiles(pa, nb, b, IT_INIT1, 0, nb->npad / TILE, false, cur);
  } else {
    p_block_done(pa, nb, st, b, err, it, ratio, tries, total_iters, sel);
  }
}

__device__ inline void p_start_step(const PArgs& pa, const NewtonBlock* nb, int b, int cur,
                                    int avg) {
  const int bits = (cur & 1) | (avg ? 2 : 0);
  release_agent();
  push_product(pa, nb, b, 0, bits);
  if (nb->nprod >= 3) push_product(pa, nb, b, 1, bits);
}

// after init2 (DS:874-877)
__device__ inline void p_control_init(const PArgs& pa, const NewtonBlock* nb, NewtonState* st,
                                      int b, int cur) {
  const float err = __uint_as_float(ald(&st->err_bits));
  ast(&st->err_bits, 0u);
  ast(&st->err, err);
  ast(&st->ratio, 1.f);
  ast(&st->it, 0);
  ast(&st->asym_bits, 0u);
  ast(&st->xmax_bits, 0u);
  const int avg0 = (ald(&st->general) == 0 && newton_avg_next(0, ald(&st->navg), ald(&st->err), pa.avg_thr)) ? 1 : 0;
  ast(&st->avg_on, avg0);
  const bool cont = 0 < pa.num_iters && err > pa.tol;
  if (cont) { ast(&st->phase, (int)PH_ACTIVE); p_start_step(pa, nb, b, cur, avg0); }
  else p_finish_try(pa, nb, st, b, err, 0, 1.f, cur, ald(&st->total_iters));
}

// after the M update of a step (DS:848 carry + DS:836-840 condition)
__device__ inline void p_control_step(const PArgs& pa, const NewtonBlock* nb, NewtonState* st,
                                      int b, int cur) {
  const float new_err = __uint_as_float(ald(&st->err_bits));
  ast(&st->err_bits, 0u);
  const float ratio = __fdiv_rn(new_err, ald(&st->err));
  const int it = ald(&st->it) + 1;
  const int total = ald(&st->total_iters) + 1;
  const int ncur = cur ^ 1;
  ast(&st->ratio, ratio);
  ast(&st->err, new_err);
  ast(&st->it, it);
  ast(&st->total_iters, total);
  ast(&st->cur, ncur);
  const unsigned ab = ald(&st->asym_bits), xb = ald(&st->xmax_bits);
  const int was = ald(&st->avg_on);
  (void)ab; (void)xb;
  if (was) ast(&st->asym_first, ald(&st->asym_first) + 1.f);
  const int avg = (was && newton_avg_next(it, ald(&st->navg), new_err, pa.avg_thr)) ? 1 : 0;
  ast(&st->avg_on, avg);
  ast(&st->asym_bits, 0u);
  ast(&st->xmax_bits, 0u);
  const bool cont = it < pa.num_iters && new_err > pa.tol && ratio < 1.2f;
  if (cont) p_start_step(pa, nb, b, ncur, avg);
  else p_finish_try(pa, nb, st, b, new_err, it, ratio, ncur, total);
}

// Arrival of one finished item (one lane, after the workgroup's drain + barrier).
__device__ inline void p_complete(const PArgs& pa, u64 item) {
  const int b = (int)((item >> 32) & 0xffffu), kind = (int)((item >> 28) & 0xfu);
  const int prod = (int)((item >> 24) & 0xfu), bits = (int)(item & 3u), bit = bits & 1;
  const NewtonBlock* nb = &pa.blocks[b];
  NewtonState* st = &pa.states[b];
  const unsigned t = (unsigned)(nb->npad / TILE);
  release_agent();
  if (kind == IT_PROD) {
    const unsigned want = t * (t + 1) / 2;
    if (__hip_atomic_fetch_add(&st->c_prod[prod], 1u, PS_RLX, PS_AGENT) + 1u != want) return;
    ast(&st->c_prod[prod], 0u);
    const int L = nb->nprod - 1;
    if (prod == L) { p_control_step(pa, nb, st, b, bit); return; }
    if (prod >= 1 && prod < L - 1) { release_agent(); push_product(pa, nb, b, prod + 1, bits); }
    if (prod == 0 || prod == L - 1) {
      const unsigned target = L >= 2 ? 2u : 1u;
      if (__hip_atomic_fetch_add(&st->c_join, 1u, PS_RLX, PS_AGENT) + 1u == target) {
        ast(&st->c_join, 0u);
        release_agent();
        push_product(pa, nb, b, L, bits);
      }
    }
  } else if (kind == IT_INIT1) {
    if (__hip_atomic_fetch_add(&st->c_init1, 1u, PS_RLX, PS_AGENT) + 1u != t * t) return;
    ast(&st->c_init1, 0u);
    release_agent();
    push_tiles(pa, nb, b, IT_INIT2, 0, (int)t, false, bit);
  } else if (kind == IT_INIT2) {
    if (__hip_atomic_fetch_add(&st->c_init2, 1u, PS_RLX, PS_AGENT) + 1u != t * t) return;
    ast(&st->c_init2, 0u);
    p_control_init(pa, nb, st, b, bit);
  } else {  // IT_COPY
    const unsigned tc = (unsigned)((nb->n_full + TILE - 1) / TILE);
    if (__hip_atomic_fetch_add(&st->c_copy, 1u, PS_RLX, PS_AGENT) + 1u != tc * tc) return;
    if (__hip_atomic_fetch_add(&pa.ctl->blocks_done[0], 1u, PS_RLX, PS_AGENT) + 1u !=
        (unsigned)pa.nlive)
      return;
    // the last block of the call: one exit token per workgroup in every queue
    for (int q = 0; q < pa.nq; ++q) {
      unsigned pos = q_reserve(pa, q, (unsigned)pa.grid);
      for (int k = 0; k < pa.grid; ++k) q_put(pa, q, pos++, make_item(0, IT_EXIT, 0, 0, 0, 0));
    }
  }
}

// Seeds the queues: the first initialisation pass of every live block.
__global__ void newton_seed_kernel(PArgs pa) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= pa.nblocks) return;
  const NewtonBlock* nb = &pa.blocks[b];
  if (nb->n < 1) return;
  push_tiles(pa, nb, b, IT_INIT1, 0, nb->npad / TILE, false, pa.states[b].cur);
}

// Payload stores of the persistent execution: plain (write-back L2, made visible by the
// agent-scope release that precedes every arrival) or write-through (sc1).  4-byte sc1
// stores are one fabric write each (MI355X_MICROARCH.md: ~6x the time per byte of a
// 16-byte one) and were measured slower here: epilogues 46 -> ? us per tile.
constexpr bool PERSIST_WT = false;
// LDS of a product workgroup: the A (k-contiguous) and B (mn-contiguous) images, double
// buffered, and at least the 64 x 129 floats of the mirror-store staging.
template <int BK>
constexpr int PSMEM = SmemCfg<BK>::TOTAL > 64 * 132 ? SmemCfg<BK>::TOTAL : 64 * 132;

// DEEP = false: 3 workgroups per CU (<= 168 VGPRs).  DEEP = true: two-K-tile-deep register
// prefetch in the K loop, 2 workgroups per CU (the second register set does not fit in 168).
template <int BK, bool DEEP, bool CAREFUL = false>
__global__ __launch_bounds__(256, (DEEP || CAREFUL) ? 2 : 3) void newton_persistent_kernel(PArgs pa) {
  extern __shared__ __align__(16) float smem[];  // PSMEM<BK> floats + 16
  float* scratch = smem + PSMEM<BK>;             // 16 floats of control scratch
  u64* s_item = reinterpret_cast<u64*>(scratch + 12);
  const int tid = threadIdx.x;
  // Queue of this workgroup.  Workgroups are dealt round-robin over the XCDs, so blockIdx % 8
  // keeps a queue's blocks in one XCD's L2 (speed only); unlike the XCC id it also
  // guarantees that every queue is served whatever the partition mode of the device.
  const int my_q = (int)(blockIdx.x % (unsigned)pa.nq);
  // dev profile (tid 0 only; 100 MHz ticks): pop, acquire, K loops, epilogues, drain,
  // completion, items, product items
  u64 pf[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  u64* s_stamp = reinterpret_cast<u64*>(scratch + 8);
  u64 t_pop_end = 0, t_work = 0;
  for (;;) {
    if (tid == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      u64 it = q_pop_wait(pa, my_q);
      if (it != ITEM_EXIT && (int)((it >> 28) & 0xfu) == IT_EXIT) it = ITEM_EXIT;
      if (pa.prof) t_pop_end = __builtin_amdgcn_s_memrealtime();
      if (it != ITEM_EXIT) {
        // ONE agent-scope acquire after the successful pop: drops this CU's stale L1 lines;
        // the wait holds the barrier below until the invalidate has completed.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (pa.prof) {
        t_work = __builtin_amdgcn_s_memrealtime();
        pf[0] += t_pop_end - t0; pf[1] += t_work - t_pop_end;
        *s_stamp = 0;
      }
      *s_item = it;
    }
    __syncthreads();
    const u64 item = *s_item;
    if (item == ITEM_EXIT) {
      if (pa.prof && tid == 0) {
        pf[10] = (u64)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);  // XCC id
        for (int k = 0; k < 12; ++k) pa.prof[(size_t)blockIdx.x * 12 + k] = pf[k];
      }
      return;
    }
    const int b = (int)((item >> 32) & 0xffffu), kind = (int)((item >> 28) & 0xfu);
    const int prod = (int)((item >> 24) & 0xfu), tm = (int)((item >> 16) & 0xffu);
    const int tn = (int)((item >> 8) & 0xffu), bit = (int)(item & 1u);
    const int avg = (int)((item >> 1) & 1u);
    const NewtonBlock* nb = &pa.blocks[b];
    NewtonState* st = &pa.states[b];
    if (kind == IT_PROD) {
      if (!(prod == 0 && ald(&st->it) == 0))   // step-0 P0: written by newton_init2_tile
        newton_product_item<BK, PERSIST_WT, DEEP, 0, false, CAREFUL>(nb, st, prod, bit, avg, tm, tn, smem,
                                      pa.prof ? s_stamp : nullptr);
    } else if (kind == IT_INIT1) {
      newton_init1_tile<PERSIST_WT>(nb, ald(&st->ridge_try), tm, tn, scratch);
    } else if (kind == IT_INIT2) {
      newton_init2_tile<PERSIST_WT>(nb, st, ald(&st->ridge_try), bit, tm, tn, scratch);
    } else {
      newton_copy_tile(nb, bit, tm, tn);
    }
    // every storing wave drains its write-through stores, then the workgroup's barrier,
    // then ONE lane signals (G16 R1)
    u64 t_epi_end = 0;
    if (pa.prof && tid == 0) t_epi_end = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      u64 t_drain_end = 0;
      if (pa.prof) t_drain_end = __builtin_amdgcn_s_memrealtime();
      p_complete(pa, item);
      if (pa.prof) {
        const u64 t_end = __builtin_amdgcn_s_memrealtime();
        const u64 t_k = *s_stamp;  // end of the (last) K loop of a product item, else 0
        if (t_k != 0) { pf[2] += t_k - t_work; pf[3] += t_epi_end - t_k; pf[7] += 1; }
        else pf[3] += t_epi_end - t_work;
        pf[4] += t_drain_end - t_epi_end; pf[5] += t_end - t_drain_end; pf[6] += 1;
      }
    }
  }
}

// All-padding blocks (zero result, error 0, DS:930-937) and, if the persistent kernel
// gave up (a bounded spin expired: must never happen), NaN errors for unfinished blocks
// so that the caller's failure select keeps the previous preconditioners.
__global__ __launch_bounds__(256) void newton_persist_epilogue_kernel(PArgs pa) {
  const int b = blockIdx.x;
  const NewtonBlock* nb = &pa.blocks[b];
  NewtonState* st = &pa.states[b];
  const int nf = nb->n_full;
  if (nb->n == 0) {
    for (int64_t e = threadIdx.x; e < (int64_t)nf * nf; e += 256)
      nb->out[(e / nf) * nb->ldo + e % nf] = 0.f;
    if (threadIdx.x == 0)
      write_metrics(pa.metrics, b, 0.f, 0, 1.f, 1, 0, st->max_ev, st->power_iters);
  } else if (st->phase != PH_DONE && threadIdx.x == 0) {
    write_metrics(pa.metrics, b, __uint_as_float(0x7fc00000u), st->it, st->ratio, st->tries,
                  st->total_iters, st->max_ev, st->power_iters);
  }
}

}  // namespace psk

// =============================================================================
// host side
// =============================================================================
using namespace psk;
using psh::Arena;

namespace {

struct Product { int a, b, c; };  // buffer ids

// Products of one Newton step for exponent p, in dependency order.  Mirrors the
// loop of mat_power (DS:663-677): power <- mat @ power on odd bits, mat <- mat @
// mat after each bit (the first odd bit just aliases power = mat, because the
// reference's initial power is the identity).  Returns false if p needs more
// temporaries than NTEMP.
bool build_chain(int p, std::vector<Product>& chain, int& power_id) {
  chain.clear();
  int next_t = 0;
  auto new_t = [&]() { return ID_T0 + next_t++; };
  int mat = ID_MI, power = -1, i = p;
  while (i > 0) {
    if (i & 1) {
      if (power < 0) power = mat;
      else { int t = new_t(); chain.push_back({mat, power, t}); power = t; }
    }
    i >>= 1;
    if (i > 0) { int t = new_t(); chain.push_back({mat, mat, t}); mat = t; }
  }
  power_id = power;
  // Temporaries are never recycled within a step: simple, and NTEMP=5 covers
  // every exponent <= 8 plus {10, 12, 16}.
  return next_t <= NTEMP && (int)chain.size() + 2 <= MAX_PROD;
}

struct Plan {
  int batch = 0;
  std::vector<int> n_eff, npad;
  std::vector<std::vector<Product>> chains;  // per block, incl. H and M products
  std::vector<int> queue_of;                 // persistent execution
  size_t qcap = 0;
  int nq = 1;
  int nlive = 0;
  // staged execution
  int nstages = 0;
  std::vector<std::vector<TileEntry>> stage_tiles;
  // the same tiles in the order used while M updates are averaged (two-pass tiles first)
  std::vector<std::vector<TileEntry>> stage_tiles_avg;
  std::vector<TileEntry> init_tiles;  // one per (block, tile)
  // stream groups (staged execution): the tile lists above are group-major; group g owns the
  // slice [goff[s][g], goff[s][g + 1]) of stage s (ioff: of the init list) and the blocks gids[g]
  int ngroups = 1;
  std::vector<std::vector<int>> goff;   // [nstages][ngroups + 1]
  std::vector<int> ioff;                // [ngroups + 1]
  std::vector<std::vector<int>> gids;   // [ngroups] block ids, ascending
  PiPlan pip;
  int max_n = 0;
  bool ok = true;
};
constexpr int MAX_GROUPS = 4;

// Which execution a call takes (read per call: tests and A/B runs flip it inside one
// process).  Default = staged: measured on MI355X (profiles/r02_*), cfg2 256 x 512^2:
// staged product launches 13.5 ms vs persistent 15.6 ms — the dataflow kernel removes the
// 24 grid drains, but every tile then pays a software dequeue + acquire + release/arrival
// (~10 us of 143 us) that the hardware dispatcher does for free, and the MFMA pipe of a CU
// stays shared by only 3 workgroups either way.  PS_NEWTON_PERSISTENT=1 selects the
// persistent kernel (no host round trip at all: the call only enqueues).
// (selected per call: ps_options.execution)

// groups: stream groups of the staged execution (newton_driver); weight (may be NULL): relative number of
// Newton steps a block is expected to take (the caller's iteration-count hint) for balancing them.
void make_plan(Plan& pl, int batch, const int32_t* n, const int32_t* p,
               const int32_t* padding_start, bool staged, int groups = 1,
               const float* weight = nullptr, int weight_stride = 1) {
  pl.batch = batch;
  pl.n_eff.resize(batch);
  pl.npad.resize(batch);
  pl.chains.resize(batch);
  pl.queue_of.assign(batch, 0);
  for (int b = 0; b < batch; ++b) {
    int ne = n[b];
    if (padding_start) ne = std::max(0, std::min(ne, (int)padding_start[b]));
    pl.n_eff[b] = ne;
    pl.npad[b] = ne >= 1 ? psh::round_up(ne, TILE) : 0;
    pl.max_n = std::max(pl.max_n, ne);
    if (ne >= 1) {
      ++pl.nlive;
      std::vector<Product> ch;
      int power_id;
      if (p[b] < 1 || !build_chain(p[b], ch, power_id)) { pl.ok = false; return; }
      // H <- H Mi first; M <- power @ M is last.
      std::vector<Product> full;
      full.push_back({ID_HCUR, ID_MI, ID_HNEXT});
      for (auto& q : ch) full.push_back(q);
      full.push_back({power_id, ID_MCUR, ID_MNEXT});
      pl.chains[b] = full;
    }
  }
  pl.pip.build(batch, pl.n_eff);

  // ---- persistent execution: block -> queue, queue capacity.  A workgroup serves ONE queue
  // (blockIdx % nq), so the queues must carry equal work: blocks are dealt by longest-
  // processing-time on c(p) * T^3 for every queue count 1..8 and the count with the smallest
  // makespan (max queue load x queue count; ties -> more queues, for L2 locality) is used.
  {
    std::vector<int> order;
    for (int b = 0; b < batch; ++b) if (pl.n_eff[b] >= 1) order.push_back(b);
    auto cost = [&](int b) {
      const double t = pl.npad[b] / (double)TILE;
      return (double)pl.chains[b].size() * t * t * t;
    };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost(x) > cost(y); });
    double best = -1.0;
    std::vector<int> assign(batch, 0);
    for (int nq = NQ; nq >= 1; --nq) {
      double load[NQ] = {0};
      for (int b : order) {
        int q = 0;
        for (int k = 1; k < nq; ++k) if (load[k] < load[q]) q = k;
        assign[b] = q;
        load[q] += cost(b);
      }
      double mx = 0;
      for (int k = 0; k < nq; ++k) mx = std::max(mx, load[k]);
      if (best < 0 || mx * nq < best * (1.0 - 1e-9)) {
        best = mx * nq;
        pl.nq = nq;
        pl.queue_of = assign;
      }
    }
    size_t cap[NQ] = {0};
    for (int b : order) {
      const size_t t = pl.npad[b] / TILE, tc = (n[b] + TILE - 1) / TILE;
      cap[pl.queue_of[b]] += std::max(2 * t * t, tc * tc);  // most items a block can have outstanding
    }
    size_t need = 64;
    for (int k = 0; k < NQ; ++k) need = std::max(need, cap[k] + 64 + 4096);  // + exit tokens
    size_t c = 64;
    while (c < need) c <<= 1;
    pl.qcap = c;
  }
  if (!staged) return;

  // ---- staged execution: stage s runs product s+1 of every block (product 0, the H
  // update, joins stage 0); the M update of a block runs as soon as its chain is done.
  // (A block whose step is only {H update, M update} (p = 1) keeps them in
  // different stages: the M update's epilogue rewrites Mi, which H reads.)
  auto stage_of = [](size_t k, size_t len) {
    if (k == 0) return 0;
    if (len == 2) return 1;
    return (int)k - 1;
  };
  pl.nstages = 0;
  for (auto& c : pl.chains)
    if (!c.empty())
      pl.nstages = std::max(pl.nstages, stage_of(c.size() - 1, c.size()) + 1);
  pl.stage_tiles.assign(pl.nstages, {});
  // Stream groups: the live blocks are dealt to `groups` groups by longest-processing-time on (upper
  // tile triangle x products per step x expected steps); every group runs the same per-block
  // arithmetic on its own stream, so results do not depend on the grouping.
  pl.ngroups = std::max(1, std::min(std::min(groups, MAX_GROUPS), std::max(1, pl.nlive)));
  pl.gids.assign(pl.ngroups, {});
  {
    std::vector<int> order;
    for (int b = 0; b < batch; ++b) if (!pl.chains[b].empty()) order.push_back(b);
    auto cost = [&](int b) {
      const double t = pl.npad[b] / TILE;
      double w = 1.0;
      if (weight) { const float h = weight[(size_t)b * weight_stride]; if (h >= 1.f && h <= 1000.f) w = h; }
      return t * (t + 1) * 0.5 * (double)pl.chains[b].size() * w * (double)pl.npad[b];
    };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost(x) > cost(y); });
    std::vector<double> load(pl.ngroups, 0.0);
    for (int b : order) {
      int g = 0;
      for (int k = 1; k < pl.ngroups; ++k) if (load[k] < load[g]) g = k;
      load[g] += cost(b);
      pl.gids[g].push_back(b);
    }
    for (auto& v : pl.gids) std::sort(v.begin(), v.end());
  }
  pl.goff.assign(pl.nstages, std::vector<int>(pl.ngroups + 1, 0));
  pl.ioff.assign(pl.ngroups + 1, 0);
  for (int g = 0; g < pl.ngroups; ++g) {
    for (int b : pl.gids[g]) {
      auto& c = pl.chains[b];
      const int t = pl.npad[b] / TILE;
      for (int tm = 0; tm < t; ++tm)
        for (int tn = 0; tn < t; ++tn) pl.init_tiles.push_back({b, (short)tm, (short)tn});
      for (size_t k = 0; k < c.size(); ++k) {
        const int s = stage_of(k, c.size());
        for (int tm = 0; tm < t; ++tm)
          for (int tn = tm; tn < t; ++tn)
            pl.stage_tiles[s].push_back({b | ((int)k << 24), (short)tm, (short)tn});
      }
    }
    pl.ioff[g + 1] = (int)pl.init_tiles.size();
    for (int s = 0; s < pl.nstages; ++s) pl.goff[s][g + 1] = (int)pl.stage_tiles[s].size();
  }
  // Launches of steps that average the M update: the off-diagonal tiles of that product run two
  // passes (TF_RAW + TF_AVG), twice as long as every other tile of the launch.  Workgroup i runs
  // list entry xcd_remap(i): XCD x walks its contiguous chunk of the list in order, so inside every
  // chunk the two-pass tiles are moved to the front (longest first: the one-pass tiles fill the
  // tail of the launch instead of waiting behind it; same tiles per XCD, so a block's operands
  // stay in one L2).
  pl.stage_tiles_avg = pl.stage_tiles;
  for (int s = 0; s < pl.nstages; ++s)
    for (int g = 0; g < pl.ngroups; ++g) {   // every group's slice is a launch of its own
      auto& v = pl.stage_tiles_avg[s];
      const int nt = pl.goff[s][g + 1] - pl.goff[s][g], q = nt >> 3, r = nt & 7;
      int first = pl.goff[s][g];
      for (int x = 0; x < 8; ++x) {
        const int cnt = x < r ? q + 1 : q;
        std::stable_partition(v.begin() + first, v.begin() + first + cnt, [&](const TileEntry& te) {
          const int b = te.task & TE_BLOCK_MASK, k = te.task >> 24;
          return k == (int)pl.chains[b].size() - 1 && te.tm != te.tn;
        });
        first += cnt;
      }
    }
}

// dev A/B: PS_NEWTON_GRID=g caps the stage launches at g workgroups that loop over the tile
// list (0 / unset: one workgroup per tile).
static int stage_grid(int ntiles, int cap) {
  return cap > 0 ? std::min(cap, ntiles) : ntiles;
}

struct WsLayout {
  NewtonBlock* blocks;
  NewtonState* states;
  PControl* ctl;
  u64* slots;
  u64* prof;
  TileEntry* tiles[MAX_PROD];
  TileEntry* tiles_avg[MAX_PROD];
  TileEntry* init_tiles;
  int* group_ids;     // [batch]: the block ids of the stream groups, group-major
  std::vector<float*> mat[10];
  std::vector<float*> sumsq;
};

size_t carve(Plan& pl, Arena& ar, WsLayout* lo, bool staged) {
  const int B = pl.batch;
  NewtonBlock* blocks = ar.take<NewtonBlock>(B);
  NewtonState* states = ar.take<NewtonState>(B);
  PControl* ctl = ar.take<PControl>(1);
  u64* slots = ar.take<u64>(NQ * pl.qcap);
  u64* prof = ar.take<u64>(12 * 4096);  // dev profile of up to 4096 workgroups
  pl.pip.carve(ar, lo != nullptr);
  if (lo) { lo->blocks = blocks; lo->states = states; lo->ctl = ctl; lo->slots = slots;
            lo->prof = prof; }
  if (staged) {
    for (int s = 0; s < pl.nstages; ++s) {
      TileEntry* e = ar.take<TileEntry>(pl.stage_tiles[s].size());
      TileEntry* e2 = ar.take<TileEntry>(pl.stage_tiles[s].size());
      if (lo) { lo->tiles[s] = e; lo->tiles_avg[s] = e2; }
    }
    TileEntry* it = ar.take<TileEntry>(pl.init_tiles.size());
    int* gi = ar.take<int>(B);
    if (lo) { lo->init_tiles = it; lo->group_ids = gi; }
  }
  for (int b = 0; b < B; ++b) {
    const size_t sq = (size_t)pl.npad[b] * pl.npad[b];
    for (int k = 0; k < 10; ++k) {
      float* m = ar.take<float>(sq);
      if (lo) lo->mat[k].push_back(m);
    }
    const int t = pl.npad[b] / TILE;
    float* ss = ar.take<float>(std::max(1, t * t));
    if (lo) lo->sumsq.push_back(ss);
  }
  return ar.off;
}

struct Profile {
  bool on = false;
  double stage_ms = 0, pi_ms = 0, other_ms = 0;
  int64_t stage_launches = 0;
};
Profile g_prof;

// Event pairs recorded on the stream the bracketed launches go to; resolved after the call.
struct ProfRun {
  struct Span { hipEvent_t a, b; int kind; int count; };  // kind 0 products, 1 power iter, 2 other
  std::vector<Span> spans;
  bool active;
  hipStream_t st;
  hipEvent_t base = nullptr;   // time origin (caller's stream): spans of different streams on one axis
  explicit ProfRun(hipStream_t s) : active(g_prof.on), st(s) {
    if (active && (hipEventCreate(&base) != hipSuccess || hipEventRecord(base, st) != hipSuccess)) active = false;
  }
  // count: launches the span brackets (kind 0).  An event record is a queue packet with a
  // completion signal: a pair around EVERY product launch kept the next kernel from starting
  // under the tail of the previous one and read 0.487 ms per launch where rocprofv3 and the
  // step time say 0.44; the product launches of a Newton step are bracketed together.
  void begin(int kind, int count = 1, hipStream_t on = nullptr) {
    if (!active) return;
    Span sp; sp.kind = kind; sp.count = count;
    if (hipEventCreate(&sp.a) != hipSuccess || hipEventCreate(&sp.b) != hipSuccess) { active = false; return; }
    (void)hipEventRecord(sp.a, on ? on : st);
    spans.push_back(sp);
  }
  void end(hipStream_t on = nullptr) {
    if (!active || spans.empty()) return;
    (void)hipEventRecord(spans.back().b, on ? on : st);
  }
  // Product spans of the stream groups overlap in time: stage_ms is the length of the UNION of their
  // intervals (the time during which at least one group had product launches in flight), so that
  // flops / stage_ms is the device's rate over the product phase whatever the grouping.
  void finish() {
    if (spans.empty()) { if (base) (void)hipEventDestroy(base); return; }
    (void)hipStreamSynchronize(st);   // the side streams were joined into st before this
    std::vector<std::pair<float, float>> prod;
    for (auto& sp : spans) {
      float ms = 0.f, t0 = 0.f;
      if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
        if (sp.kind == 0) {
          g_prof.stage_launches += sp.count;
          if (hipEventElapsedTime(&t0, base, sp.a) == hipSuccess) prod.push_back({t0, t0 + ms});
          else g_prof.stage_ms += ms;
        }
        else if (sp.kind == 1) g_prof.pi_ms += ms;
        else g_prof.other_ms += ms;
      }
      (void)hipEventDestroy(sp.a);
      (void)hipEventDestroy(sp.b);
    }
    std::sort(prod.begin(), prod.end());
    float cur_a = 0.f, cur_b = -1.f;
    for (auto& iv : prod) {
      if (cur_b < cur_a) { cur_a = iv.first; cur_b = iv.second; }
      else if (iv.first <= cur_b) cur_b = std::max(cur_b, iv.second);
      else { g_prof.stage_ms += cur_b - cur_a; cur_a = iv.first; cur_b = iv.second; }
    }
    if (cur_b >= cur_a && !prod.empty()) g_prof.stage_ms += cur_b - cur_a;
    spans.clear();
    if (base) { (void)hipEventDestroy(base); base = nullptr; }
  }
};

// Side streams of the stream groups: the library's shared pool (common.h psh::side_stream; one device per
// process: PS_DEVICE_CHECK).
inline hipStream_t side_stream(int k) { return psh::side_stream(k); }

// One mapped status ring per host thread: a thread runs one call at a time, so calls
// on distinct (stream, workspace) pairs from different threads never share slots.
HostStatus* pinned_status() {
  static thread_local HostStatus* st = nullptr;
  if (!st) {
    if (hipHostMalloc((void**)&st, 64 * MAX_GROUPS * sizeof(HostStatus), hipHostMallocMapped) !=
        hipSuccess)
      st = nullptr;
  }
  return st;
}

// Resident grid of the persistent kernel (workgroups per CU from the occupancy query x
// CUs).  Correctness does not depend on residency (no workgroup ever waits for a specific
// other workgroup: waits are for queue items, which only running workgroups produce),
// so the query is for speed only.
// Deep-prefetch / 2-per-CU variant of the product kernels.  Staged execution: on by default
// with BK = 32 (measured, 256 x 512^2 / 64 x 1024^2 product launches: BK16 13.75 / 22.69 ms,
// BK16+deep 13.73 / 22.39, BK32 14.51 / 22.76, BK32+deep 13.57 / 21.70).  Persistent execution:
// off by default (3 workgroups per CU cover each other's dequeue / epilogue phases better).
// PS_NEWTON_DEEP = 0 | 1 and PS_NEWTON_BK = 16 | 32 override.
int persistent_grid(size_t lds_bytes, bool deep, int wg_per_cu) {
  static int grid[2] = {0, 0};
  if (grid[deep] == 0) {
    int dev = 0, cus = 256, occ = deep ? 2 : 3;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    const void* fn = deep ? (const void*)newton_persistent_kernel<NBK, true>
                          : (const void*)newton_persistent_kernel<NBK, false>;
    int q = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, fn, 256, lds_bytes) == hipSuccess && q >= 1)
      occ = q;
    if (wg_per_cu > 0) occ = wg_per_cu;
    grid[deep] = cus * occ;
  }
  return grid[deep];
}

}  // namespace

extern "C" int ps_newton_averaged_steps(void) { return psh::resolve(nullptr).averaged_steps; }

extern "C" size_t ps_newton_root_workspace_bytes(int batch, const int32_t* n,
                                                 const int32_t* p,
                                                 const int32_t* padding_start) {
  if (batch <= 0 || !n || !p) return 0;
  Plan pl;
  make_plan(pl, batch, n, p, padding_start, true);  // staged tables included: upper bound
  if (!pl.ok) return 0;
  Arena ar(nullptr, 0);
  return carve(pl, ar, nullptr, true) + 256;
}

static int newton_driver(
    void* stream, const float* const* a, const int32_t* n, const int32_t* lda,
    const int32_t* p, const int32_t* padding_start, int batch, int num_iters,
    float ridge_epsilon, float error_tolerance, int relative_matrix_epsilon,
    const float* max_ev_given, int symmetry, float* const* out, const int32_t* ldo,
    float* metrics, void* workspace, size_t workspace_bytes, int32_t* iters_executed_host,
    const ps_options* options) {
  PS_DEVICE_CHECK();
  bool bad_options = false;
  const psh::Options opt = psh::resolve(options, &bad_options);
  if (bad_options) return PS_EINVAL;
  if (batch <= 0 || !a || !n || !lda || !p || !out || !ldo || !metrics || !workspace ||
      num_iters < 1 || symmetry < PS_SYMMETRY_VERIFY || symmetry > PS_SYMMETRY_GENERAL)
    return PS_EINVAL;
  if (batch > 65535) return PS_EUNSUPPORTED;  // 16-bit block field of a queue item
  for (int b = 0; b < batch; ++b)
    if (n[b] < 1 || lda[b] < n[b] || ldo[b] < n[b] || !a[b] || !out[b]) return PS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool staged = opt.execution != PS_EXEC_PERSISTENT;
  if (opt.force_general) symmetry = PS_SYMMETRY_GENERAL;   // dev
  // Leading steps of every try whose M update is computed in full and averaged with its
  // transpose (TileFlags above); per block 0 when the caller's hint marks it well conditioned.
  const int navg = opt.averaged_steps;
  const float avg_thr = opt.averaged_err_threshold;
  const int seg_on = (opt.accumulation == PS_ACCUM_SEGMENTED && opt.products == PS_PRODUCTS_F32) ? 1 : 0;
  Plan pl;
  // Stream groups of the staged execution (see the product loop below): 2 by default when the call has
  // work for them (PS_NEWTON_GROUPS / psh::Options::newton_groups: 1 = the single-stream execution).
  // Measured (tools/dev_stage_variants.py groups, profiles/r06_newton_stream_groups.txt): calls whose
  // blocks differ in size or exponent (a ViT-B tree: 139.6 -> 133.9 ms with 2 groups, 135-136 with 3 / 4)
  // gain; homogeneous batches do not (256 x 512^2: 16.1 vs 16.2-16.3 ms, 64 x 1024^2: 23.5 vs 23.7 -- their
  // launches are whole rounds of equal tiles and the second stream only costs L2 locality), so the default
  // is 2 groups for mixed calls and 1 otherwise.
  int want_groups = opt.newton_groups;
  if (want_groups <= 0) {
    want_groups = 1;
    int n0 = -1, p0 = -1;
    for (int b = 0; b < batch; ++b) {
      int ne = n[b];
      if (padding_start) ne = std::max(0, std::min(ne, (int)padding_start[b]));
      if (ne < 1) continue;
      const int np_ = psh::round_up(ne, TILE);
      if (n0 < 0) { n0 = np_; p0 = p[b]; }
      else if (np_ != n0 || p[b] != p0) { want_groups = 2; break; }
    }
  }
  if (!staged || opt.newton_trace) want_groups = 1;
  make_plan(pl, batch, n, p, padding_start, staged, want_groups, opt.iters_hint, opt.iters_hint_stride);
  if (!pl.ok) return PS_EUNSUPPORTED;
  if (pl.max_n > 16384) return PS_EUNSUPPORTED;
  // De-phasing of co-resident workgroups (newton_stage_kernel): measured 64 x 1024^2: 21.6 -> 20.9 ms
  // (-2 ... -5 % by box), no effect at 512^2 (a tile is too short for the delay to pay) or on the
  // mixed tile lengths of a ViT-B launch: on by itself only when every tile of the call has K >= 1024.
  int min_npad = 1 << 30;
  for (int b = 0; b < batch; ++b)
    if (!pl.chains[b].empty()) min_npad = std::min(min_npad, pl.npad[b]);
  const int stagger = opt.stagger >= 0 ? opt.stagger : (min_npad >= 1024 && min_npad < (1 << 30) ? ((2 << 16) | 25) : 0);
  pl.pip.set_options(opt);
  Arena ar(workspace, workspace_bytes);
  WsLayout lo;
  carve(pl, ar, &lo, staged);
  if (ar.overflow) return PS_EWORKSPACE;

  // ---- upload plan (pinned staging ring: no stream synchronisation) ------------------
  std::vector<NewtonBlock> hb(batch);
  std::vector<NewtonState> hs(batch);
  bool any_careful = false, any_avg = false;
  for (int b = 0; b < batch; ++b) {
    NewtonBlock& nb = hb[b];
    NewtonState& ns = hs[b];
    memset(&nb, 0, sizeof(nb));
    memset(&ns, 0, sizeof(ns));
    nb.a = a[b]; nb.out = out[b];
    nb.M[0] = lo.mat[0][b]; nb.M[1] = lo.mat[1][b];
    nb.H[0] = lo.mat[2][b]; nb.H[1] = lo.mat[3][b];
    nb.Mi = lo.mat[4][b];
    for (int k = 0; k < NTEMP; ++k) nb.T[k] = lo.mat[5 + k][b];
    nb.sumsq_partial = lo.sumsq[b];
    nb.n = pl.n_eff[b]; nb.n_full = n[b]; nb.lda = lda[b]; nb.ldo = ldo[b];
    nb.npad = pl.npad[b]; nb.p = p[b];
    nb.alpha = (float)(-1.0 / p[b]);        // DS:774 (float32 of the exact quotient)
    nb.one_minus_alpha = 1.0f - nb.alpha;   // DS:844, float32 subtraction
    nb.inv_p = (float)(1.0 / p[b]);
    nb.nprod = (int)pl.chains[b].size();
    nb.queue = pl.queue_of[b];
    for (size_t k = 0; k < pl.chains[b].size(); ++k) {
      nb.pa[k] = (short)pl.chains[b][k].a;
      nb.pb[k] = (short)pl.chains[b][k].b;
      nb.pc[k] = (short)pl.chains[b][k].c;
    }
    ns.phase = pl.n_eff[b] >= 1 ? PH_INIT : PH_DONE;
    ns.ratio = 1.f;
    // A block whose hint (its Newton iteration count at the previous recompute) lies in
    // [1, fast_max_iters] is well conditioned: mirrored M updates are exact to 1e-6 there.  Every
    // other block (no hint, NaN, a slow block) averages the M update of its first steps.
    const float hv = opt.iters_hint ? opt.iters_hint[(size_t)b * opt.iters_hint_stride] : 0.f;
    const bool fast = hv >= 1.f && hv <= (float)opt.fast_max_iters;   // false for NaN
    ns.navg = fast ? 0 : navg;
    // the accumulation mode is the call's, not the block's: a block's arithmetic does not depend
    // on its hint beyond the averaged steps
    ns.seg = seg_on;
    if (pl.n_eff[b] >= 1) { any_careful |= ns.seg != 0; any_avg |= ns.navg > 0; }
  }
  ProfRun prof(st);
  // Everything from the descriptor upload to the seeded loop state; run again (on the streaming
  // power iteration) if the resident power iteration reports an expired wait.
  const unsigned pi_expired_before = PiPlan::expired_total();
  auto enqueue_front = [&]() -> int {
    PS_RC(pl.pip.upload(st, a, lda));  // also fixes pl.pip.d_asym
    for (int b = 0; b < batch; ++b) hb[b].asym = pl.pip.d_asym + b;
    PS_RC(psh::upload_async(st, lo.blocks, hb.data(), sizeof(NewtonBlock) * batch));
    PS_RC(psh::upload_async(st, lo.states, hs.data(), sizeof(NewtonState) * batch));
    PS_RC(pl.pip.enqueue_symmetry(st, symmetry));
    // ---- power iteration -> ridge epsilon --------------------------------------
    prof.begin(1);
    if (relative_matrix_epsilon && !max_ev_given) {
      int rc = pl.pip.enqueue(st, 100, 1e-6f);  // DS:820-825
      if (rc) return rc;
    }
    prof.end();
    hipLaunchKernelGGL(newton_setup_kernel, dim3((batch + 255) / 256), dim3(256), 0, st,
                       lo.states, pl.pip.d_blocks, batch, ridge_epsilon,
                       relative_matrix_epsilon, max_ev_given);
    PS_LAUNCH_CHECK();
    return 0;
  };
  if (staged) {
    for (int s = 0; s < pl.nstages; ++s) {
      PS_RC(psh::upload_async(st, lo.tiles[s], pl.stage_tiles[s].data(),
                              sizeof(TileEntry) * pl.stage_tiles[s].size()));
      if (any_avg && opt.avg_lpt)
        PS_RC(psh::upload_async(st, lo.tiles_avg[s], pl.stage_tiles_avg[s].data(),
                                sizeof(TileEntry) * pl.stage_tiles_avg[s].size()));
    }
    PS_RC(psh::upload_async(st, lo.init_tiles, pl.init_tiles.data(),
                            sizeof(TileEntry) * pl.init_tiles.size()));
    std::vector<int> ids;
    for (auto& v : pl.gids) ids.insert(ids.end(), v.begin(), v.end());
    PS_RC(psh::upload_async(st, lo.group_ids, ids.data(), sizeof(int) * ids.size()));
  }
  PS_RC(enqueue_front());

  if (!staged) {
    // ---- persistent dataflow execution: one launch, no host round trip ------------
    PArgs pa;
    pa.blocks = lo.blocks; pa.states = lo.states; pa.ctl = lo.ctl; pa.slots = lo.slots;
    pa.metrics = metrics; pa.qcap = (unsigned)pl.qcap; pa.nblocks = batch;
    pa.nlive = pl.nlive; pa.num_iters = num_iters; pa.tol = error_tolerance;
    const size_t lds = (PSMEM<NBK> + 16) * sizeof(float);
    const bool deep = opt.persistent_deep != 0;
    const int grid = std::min(persistent_grid(lds, deep || any_careful, opt.wg_per_cu), 4096);
    const bool dev_prof = opt.newton_prof;
    pa.prof = dev_prof ? lo.prof : nullptr;
    pa.grid = grid;
    pa.avg_thr = avg_thr;
    pa.nq = pl.nq;
    if (dev_prof) PS_HIP(hipMemsetAsync(lo.prof, 0, sizeof(u64) * 12 * 4096, st));
    if (pl.nlive > 0) {
      prof.begin(2);
      // every polled word (heads, tails, counters, slot tags) is zeroed before EVERY launch
      PS_HIP(hipMemsetAsync(lo.ctl, 0, sizeof(PControl), st));
      PS_HIP(hipMemsetAsync(lo.slots, 0, sizeof(u64) * NQ * pl.qcap, st));
      hipLaunchKernelGGL(newton_seed_kernel, dim3((batch + 63) / 64), dim3(64), 0, st, pa);
      prof.end();
      PS_LAUNCH_CHECK();
      prof.begin(0);
      if (any_careful)
        hipLaunchKernelGGL((newton_persistent_kernel<NBK, false, true>), dim3(grid), dim3(256), lds, st, pa);
      else if (deep)
        hipLaunchKernelGGL((newton_persistent_kernel<NBK, true>), dim3(grid), dim3(256), lds, st, pa);
      else
        hipLaunchKernelGGL((newton_persistent_kernel<NBK, false>), dim3(grid), dim3(256), lds, st, pa);
      prof.end();
      PS_LAUNCH_CHECK();
    }
    prof.begin(2);
    hipLaunchKernelGGL(newton_persist_epilogue_kernel, dim3(batch), dim3(256), 0, st, pa);
    prof.end();
    PS_LAUNCH_CHECK();
    prof.finish();
    if (dev_prof && pl.nlive > 0) {  // dev only: per-workgroup time split of the persistent kernel
      std::vector<u64> h(12 * (size_t)grid);
      PS_HIP(hipStreamSynchronize(st));
      PS_HIP(hipMemcpy(h.data(), lo.prof, sizeof(u64) * h.size(), hipMemcpyDeviceToHost));
      double sum[12] = {0};
      int per_xcd[8] = {0};
      for (int w = 0; w < grid; ++w) {
        for (int k = 0; k < 10; ++k) sum[k] += (double)h[12 * w + k];
        per_xcd[h[12 * w + 10] & 7] += 1;
      }
      const double us = 0.01 / grid;  // 100 MHz ticks -> microseconds, mean per workgroup
      fprintf(stderr, "[newton persistent] grid %d  per workgroup: items %.1f (products %.1f, stolen %.1f, "
              "empty sweeps %.1f)  pop %.0f us  acquire %.0f us  K-loops %.0f us  epilogues+other bodies "
              "%.0f us  drain %.0f us  completion %.0f us  | workgroups per XCC id: %d %d %d %d %d %d %d %d\n",
              grid, sum[6] / grid, sum[7] / grid, sum[8] / grid, sum[9] / grid, sum[0] * us, sum[1] * us,
              sum[2] * us, sum[3] * us, sum[4] * us, sum[5] * us, per_xcd[0], per_xcd[1], per_xcd[2],
              per_xcd[3], per_xcd[4], per_xcd[5], per_xcd[6], per_xcd[7]);
    }
    if (iters_executed_host) *iters_executed_host = -1;
    return PS_OK;
  }

  // ---- staged execution --------------------------------------------------------------
  HostStatus* status = pinned_status();
  if (!status) return PS_EINTERNAL;
  // Variant of the product kernel.  Default: BK = 32, two register sets of global loads, the
  // explicitly software-pipelined K loop (gemm_core.hip.h deep_run_pipe), 2 workgroups per CU;
  // the other instantiations are developer A/B variants (psh::Options::stage_bk / stage_deep /
  // pipe).  The bf16-split arithmetic has its own instantiations.
  const int stage_bk = opt.stage_bk, stage_deep = opt.stage_deep;
  const int xmode = opt.products == PS_PRODUCTS_BF16X6 ? 6 : (opt.products == PS_PRODUCTS_BF16X3 ? 3 : 0);
  const bool pipe_mode = opt.pipe != 0 && stage_bk == 32 && stage_deep;
  {
    static std::once_flag attr_once;   // kernel attributes are per process (one device per process)
    std::call_once(attr_once, [] {
      const int lds = (int)(SmemCfg<32>::TOTAL * sizeof(float));
      const void* fns[] = {(const void*)newton_stage_kernel<32, false>,
                           (const void*)newton_stage_kernel<32, true>,
                           (const void*)newton_stage_kernel<32, true, 6>,
                           (const void*)newton_stage_kernel<32, true, 3>,
                           (const void*)newton_stage_kernel<32, true, 0, false, true>,
                           (const void*)newton_stage_kernel<32, true, 0, false, true, true>,
                           (const void*)newton_stage_kernel<32, true, 0, true, true>,
                           (const void*)newton_stage_kernel<32, true, 0, true>};
      for (const void* f : fns)
        (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
  }
  // dev trace of the product kernel (PS_NEWTON_TRACE=<file>: records appended per call)
  constexpr unsigned TRACE_CAP = 1u << 20;
  static unsigned long long* trace_buf = nullptr;
  static unsigned trace_seq = 0;
  const char* trace_path = opt.newton_trace;
  if (trace_path && !trace_buf) {
    if (hipMalloc(&trace_buf, (8 + 8 * (size_t)TRACE_CAP) * sizeof(unsigned long long)) != hipSuccess)
      trace_buf = nullptr;
  }
  if (!trace_path) { /* tracing off for this call */ }
  unsigned long long* const trace_on = trace_path ? trace_buf : nullptr;
  if (trace_on) (void)hipMemsetAsync(trace_on, 0, 64, st);
  const int ninit = (int)pl.init_tiles.size();
  int executed = 0;
  if (ninit > 0) {
    // ---- stream groups -----------------------------------------------------------------------
    // The blocks are dealt to G groups (make_plan); every group runs the reference's loop for its
    // blocks -- the same launches, tile for tile, as the single-stream execution -- on a stream of
    // its own: group 0 on the caller's, the others on side streams forked behind the power iteration
    // and joined in front of the copy-out.  A stage launch depends on the previous one of ITS group
    // only, so while one group's launch drains (the last, partly filled round of 512 resident
    // workgroups; the burst of epilogue stores; the first HBM round trip of the next launch) the
    // other group's tiles keep the CUs busy.  Per-block arithmetic, and therefore every bit of the
    // result, does not depend on G.  One host event wait per group and step, one step behind the GPU.
    const int G = pl.ngroups;
    struct Grp {
      hipStream_t st = nullptr;
      hipEvent_t ev[2] = {nullptr, nullptr};
      hipEvent_t join = nullptr;
      bool need_init = true, done = false;
      int since_init = 0, steps = 0, nblk = 0, ninit = 0;
      const int* ids = nullptr;
      const TileEntry* init_tiles = nullptr;
      HostStatus* ring = nullptr;
    };
    Grp grp[MAX_GROUPS];
    hipEvent_t fork_ev = nullptr;
    int rc = 0;
    auto cleanup = [&]() {
      for (int q = 0; q < G; ++q) {
        for (int k = 0; k < 2; ++k) if (grp[q].ev[k]) (void)hipEventDestroy(grp[q].ev[k]);
        if (grp[q].join) (void)hipEventDestroy(grp[q].join);
      }
      if (fork_ev) (void)hipEventDestroy(fork_ev);
    };
    {
      int off = 0;
      for (int q = 0; q < G && !rc; ++q) {
        Grp& gr = grp[q];
        gr.st = q == 0 ? st : side_stream(q - 1);   // (the caller's stream may be the null stream)
        if (q > 0 && !gr.st) { rc = PS_EINTERNAL; break; }
        gr.nblk = (int)pl.gids[q].size();
        gr.ids = lo.group_ids + off;
        off += gr.nblk;
        gr.ninit = pl.ioff[q + 1] - pl.ioff[q];
        gr.init_tiles = lo.init_tiles + pl.ioff[q];
        gr.ring = status + 64 * q;
        for (int k = 0; k < 2 && !rc; ++k) rc = (int)hipEventCreateWithFlags(&gr.ev[k], hipEventDisableTiming);
        if (!rc && q > 0) rc = (int)hipEventCreateWithFlags(&gr.join, hipEventDisableTiming);
      }
      if (!rc && G > 1) rc = (int)hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming);
    }
    auto fork = [&]() -> int {   // the side streams start behind everything queued on st so far
      if (G == 1) return 0;
      PS_HIP(hipEventRecord(fork_ev, st));
      for (int q = 1; q < G; ++q) PS_HIP(hipStreamWaitEvent(grp[q].st, fork_ev, 0));
      return 0;
    };
    if (!rc) rc = fork();
    const int cap = 6 * (num_iters + 2) + 4;
    bool pi_retried = false;
    for (int g = 0; g < cap && !rc; ++g) {
      for (int q = 0; q < G && !rc; ++q) {
        Grp& gr = grp[q];
        if (gr.done) continue;
        HostStatus* slot = &gr.ring[g % 64];
        slot->gen = -1;
        if (gr.need_init) {
          prof.begin(2, 1, gr.st);
          hipLaunchKernelGGL(newton_init1_kernel, dim3(gr.ninit), dim3(256), 0, gr.st, lo.blocks,
                             lo.states, gr.init_tiles);
          hipLaunchKernelGGL(newton_init2_kernel, dim3(gr.ninit), dim3(256), 0, gr.st, lo.blocks,
                             lo.states, gr.init_tiles);
          hipLaunchKernelGGL(newton_control_kernel, dim3(1), dim3(256), 0, gr.st, lo.states,
                             gr.nblk, 0, num_iters, error_tolerance, g, (HostStatus*)nullptr, avg_thr, gr.ids);
          prof.end(gr.st);
          gr.since_init = 0;
        }
        const bool avg_order = any_avg && opt.avg_lpt && gr.since_init < navg;
        ++gr.since_init;
        prof.begin(0, pl.nstages, gr.st);
        for (int s = 0; s < pl.nstages; ++s) {
          const int nt = pl.goff[s][q + 1] - pl.goff[s][q];
          if (nt == 0) continue;
          const TileEntry* tl = (avg_order ? lo.tiles_avg[s] : lo.tiles[s]) + pl.goff[s][q];
          const dim3 grid(stage_grid(nt, opt.grid_cap));
          const size_t lds = SmemCfg<32>::TOTAL * sizeof(float);
          hipStream_t gst = gr.st;
#define PS_STAGE(...)                                                                        \
  hipLaunchKernelGGL((newton_stage_kernel<__VA_ARGS__>), grid, dim3(256), lds, gst, lo.blocks, \
                     lo.states, tl, nt, stagger)
          if (pipe_mode && xmode == 0 && trace_on)
            hipLaunchKernelGGL((newton_stage_kernel<32, true, 0, true, true>), grid, dim3(256), lds, gst,
                               lo.blocks, lo.states, tl, nt, navg, trace_on, trace_seq++, TRACE_CAP);
          else if (pipe_mode && xmode == 0 && any_careful) PS_STAGE(32, true, 0, false, true, true);
          else if (pipe_mode && xmode == 0) PS_STAGE(32, true, 0, false, true);
          else if (trace_on)
            hipLaunchKernelGGL((newton_stage_kernel<32, true, 0, true>), grid, dim3(256), lds, gst,
                               lo.blocks, lo.states, tl, nt, navg, trace_on, trace_seq++, TRACE_CAP);
          else if (xmode == 6) PS_STAGE(32, true, 6);
          else if (xmode == 3) PS_STAGE(32, true, 3);
          else if (stage_bk == 32 && stage_deep) PS_STAGE(32, true);
          else if (stage_bk == 32) PS_STAGE(32, false);
          else if (stage_deep)
            hipLaunchKernelGGL((newton_stage_kernel<16, true>), grid, dim3(256),
                               SmemCfg<16>::TOTAL * sizeof(float), gst, lo.blocks, lo.states, tl, nt, navg);
          else
            hipLaunchKernelGGL((newton_stage_kernel<16, false>), grid, dim3(256),
                               SmemCfg<16>::TOTAL * sizeof(float), gst, lo.blocks, lo.states, tl, nt, navg);
#undef PS_STAGE
        }
        prof.end(gr.st);
        hipLaunchKernelGGL(newton_control_kernel, dim3(1), dim3(256), 0, gr.st, lo.states,
                           gr.nblk, 1, num_iters, error_tolerance, g, slot, avg_thr, gr.ids);
        if ((rc = (int)hipGetLastError()) != 0) break;
        if ((rc = (int)hipEventRecord(gr.ev[g & 1], gr.st)) != 0) break;
        ++gr.steps;
      }
      if (rc) break;
      if (g == 0) {
        for (int q = 0; q < G; ++q) grp[q].need_init = false;
        continue;
      }
      bool restart = false;
      for (int q = 0; q < G && !rc; ++q) {
        Grp& gr = grp[q];
        if (gr.done) continue;
        if ((rc = (int)hipEventSynchronize(gr.ev[(g - 1) & 1])) != 0) break;
        const HostStatus seen = gr.ring[(g - 1) % 64];
        if (seen.gen != g - 1) { rc = PS_EINTERNAL; break; }
        if (g == 1 && !pi_retried && PiPlan::expired_total() != pi_expired_before) { restart = true; break; }
        if (seen.not_done == 0) gr.done = true;
        gr.need_init = seen.need_init > 0;
      }
      if (rc) break;
      if (restart) {
        pi_retried = true;
        // The resident power iteration gave up on a team mate that never became resident
        // (CUs held by another stream's kernels): its blocks carry a NaN eigenvalue.  The
        // process is on the streaming execution from now on (PiPlan::resident_enabled);
        // queue the whole call again behind what is already queued and start over.
        for (int q = 1; q < G && !rc; ++q) {   // join the side streams first: the front runs on st
          if ((rc = (int)hipEventRecord(grp[q].join, grp[q].st)) != 0) break;
          rc = (int)hipStreamWaitEvent(st, grp[q].join, 0);
        }
        if (rc) break;
        pl.pip.allow_resident = false;
        if ((rc = enqueue_front()) != 0) break;
        if ((rc = fork()) != 0) break;
        for (int q = 0; q < G; ++q) { grp[q].need_init = true; grp[q].done = false; grp[q].steps = 0; }
        g = -1;
        continue;
      }
      bool all_done = true;
      for (int q = 0; q < G; ++q) all_done &= grp[q].done;
      if (all_done) break;
    }
    // join: the copy-out (and whatever the caller queues next on st) runs behind every group
    for (int q = 1; q < G && !rc; ++q) {
      if ((rc = (int)hipEventRecord(grp[q].join, grp[q].st)) != 0) break;
      rc = (int)hipStreamWaitEvent(st, grp[q].join, 0);
    }
    for (int q = 0; q < G; ++q) executed = std::max(executed, grp[q].steps);
    cleanup();
    if (rc) return rc;
  }
  prof.begin(2);
  hipLaunchKernelGGL(newton_final_kernel, dim3(batch, 32), dim3(256), 0, st, lo.blocks,
                     lo.states, metrics);
  prof.end();
  PS_LAUNCH_CHECK();
  prof.finish();
  if (trace_on) {   // dev only: waits for the stream and appends this call's records to the file
    PS_HIP(hipStreamSynchronize(st));
    unsigned long long head[8];
    PS_HIP(hipMemcpy(head, trace_on, sizeof(head), hipMemcpyDeviceToHost));
    const size_t nrec = std::min<size_t>((unsigned)head[0], TRACE_CAP);
    std::vector<unsigned long long> rec(8 * nrec);
    if (nrec) PS_HIP(hipMemcpy(rec.data(), trace_on + 8, rec.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(trace_path, "ab")) { fwrite(rec.data(), 8, rec.size(), f); fclose(f); }
  }
  if (iters_executed_host) *iters_executed_host = executed;
  return PS_OK;
}

extern "C" int ps_profile_enable(int on) { g_prof.on = on != 0; return PS_OK; }
extern "C" int ps_profile_reset(void) {
  g_prof.stage_ms = g_prof.pi_ms = g_prof.other_ms = 0; g_prof.stage_launches = 0;
  return PS_OK;
}
extern "C" int ps_profile_get(double* stage_ms, int64_t* stage_launches,
                              double* power_iter_ms, double* other_ms) {
  if (stage_ms) *stage_ms = g_prof.stage_ms;
  if (stage_launches) *stage_launches = g_prof.stage_launches;
  if (power_iter_ms) *power_iter_ms = g_prof.pi_ms;
  if (other_ms) *other_ms = g_prof.other_ms;
  return PS_OK;
}

// ---- standalone power iteration ---------------------------------------------------
namespace {
void pi_plan_from_args(PiPlan& pp, int batch, const int32_t* n, const int32_t* padding_start) {
  std::vector<int> ne(batch);
  for (int b = 0; b < batch; ++b) {
    int e = n[b];
    if (padding_start) e = std::max(0, std::min(e, (int)padding_start[b]));
    ne[b] = e;
  }
  pp.build(batch, ne);
}
}  // namespace

extern "C" size_t ps_power_iteration_workspace_bytes(int batch, const int32_t* n) {
  if (batch <= 0 || !n) return 0;
  PiPlan pp;
  pi_plan_from_args(pp, batch, n, nullptr);
  Arena ar(nullptr, 0);
  pp.carve(ar, false);
  return ar.off + 256;
}

extern "C" int ps_power_iteration_batched_f32(
    void* stream, const float* const* a, const int32_t* n, const int32_t* lda,
    const int32_t* padding_start, int batch, int num_iters, float error_tolerance,
    float* out_lambda, int32_t* out_iters, float* out_v, int32_t ldv, int symmetry,
    void* workspace, size_t workspace_bytes) {
  return ps_power_iteration_batched_opt_f32(stream, a, n, lda, padding_start, batch, num_iters,
                                            error_tolerance, out_lambda, out_iters, out_v, ldv,
                                            symmetry, workspace, workspace_bytes, nullptr);
}

extern "C" int ps_power_iteration_batched_opt_f32(
    void* stream, const float* const* a, const int32_t* n, const int32_t* lda,
    const int32_t* padding_start, int batch, int num_iters, float error_tolerance,
    float* out_lambda, int32_t* out_iters, float* out_v, int32_t ldv, int symmetry,
    void* workspace, size_t workspace_bytes, const ps_options* options) {
  PS_DEVICE_CHECK();
  bool bad_options = false;
  const psh::Options opt = psh::resolve(options, &bad_options);
  if (bad_options) return PS_EINVAL;
  if (batch <= 0 || !a || !n || !lda || !out_lambda || !workspace || num_iters < 1 ||
      symmetry < PS_SYMMETRY_VERIFY || symmetry > PS_SYMMETRY_GENERAL)
    return PS_EINVAL;
  for (int b = 0; b < batch; ++b)
    if (n[b] < 1 || lda[b] < n[b] || !a[b]) return PS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  PiPlan pp;
  pi_plan_from_args(pp, batch, n, padding_start);
  pp.set_options(opt);
  if (pp.max_n > 16384) return PS_EUNSUPPORTED;
  if (out_v && ldv < pp.max_n) return PS_EINVAL;
  Arena ar(workspace, workspace_bytes);
  pp.carve(ar, true);
  if (ar.overflow) return PS_EWORKSPACE;
  int rc = pp.upload(st, a, lda);
  if (rc) return rc;
  rc = pp.enqueue_symmetry(st, symmetry);
  if (rc) return rc;
  rc = pp.enqueue(st, num_iters, error_tolerance);
  if (rc) return rc;
  hipLaunchKernelGGL(pi_output_kernel, dim3(batch), dim3(256), 0, st, pp.d_blocks, batch,
                     out_lambda, (int*)out_iters, out_v, (int)ldv);
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_newton_root_batched_f32(
    void* stream, const float* const* a, const int32_t* n, const int32_t* lda,
    const int32_t* p, const int32_t* padding_start, int batch, int num_iters,
    float ridge_epsilon, float error_tolerance, int relative_matrix_epsilon, int symmetry,
    float* const* out, const int32_t* ldo, float* metrics, void* workspace,
    size_t workspace_bytes, int32_t* iters_executed_host) {
  return newton_driver(stream, a, n, lda, p, padding_start, batch, num_iters, ridge_epsilon,
                       error_tolerance, relative_matrix_epsilon, nullptr, symmetry, out, ldo,
                       metrics, workspace, workspace_bytes, iters_executed_host, nullptr);
}

extern "C" int ps_newton_root_batched_opt_f32(
    void* stream, const float* const* a, const int32_t* n, const int32_t* lda,
    const int32_t* p, const int32_t* padding_start, int batch, int num_iters,
    float ridge_epsilon, float error_tolerance, int relative_matrix_epsilon, const float* max_ev,
    int symmetry, float* const* out, const int32_t* ldo, float* metrics, void* workspace,
    size_t workspace_bytes, int32_t* iters_executed_host, const ps_options* options) {
  return newton_driver(stream, a, n, lda, p, padding_start, batch, num_iters, ridge_epsilon,
                       error_tolerance, max_ev ? 1 : relative_matrix_epsilon, max_ev, symmetry, out,
                       ldo, metrics, workspace, workspace_bytes, iters_executed_host, options);
}

extern "C" int ps_newton_root_batched_maxev_f32(
    void* stream, const float* const* a, const int32_t* n, const int32_t* lda,
    const int32_t* p, const int32_t* padding_start, int batch, int num_iters,
    float ridge_epsilon, float error_tolerance, const float* max_ev, int symmetry,
    float* const* out, const int32_t* ldo, float* metrics, void* workspace,
    size_t workspace_bytes, int32_t* iters_executed_host) {
  if (!max_ev) return PS_EINVAL;
  return newton_driver(stream, a, n, lda, p, padding_start, batch, num_iters, ridge_epsilon,
                       error_tolerance, 1, max_ev, symmetry, out, ldo, metrics, workspace,
                       workspace_bytes, iters_executed_host, nullptr);
}
