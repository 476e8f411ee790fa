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
// Per host iteration the stream receives, for the whole batch at once:
//   stage 0..S-1 : newton_stage_kernel — one 128x128 output tile per workgroup,
//                  products of the binary-powering chain of mat_power (DS:655-678,
//                  same multiplication order, minus the exact "@ I" and the unused
//                  trailing square), H <- H Mi (DS:846) in stage 0, and as the last
//                  product of each block M <- Mi^p M (DS:845) whose epilogue also
//                  writes the next Mi (DS:844) and max|M - I| (DS:847);
//   control      : one thread per block advances the loop state of DS:836-848 /
//                  DS:858-885 (ratio guard, retry with ridge*10^i) on the device.
// Finished blocks cost nothing (their tiles exit at once).  The host only polls
// "how many blocks are still running" one iteration behind the GPU.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "gemm_core.hip.h"
#include "power_iter.hip.h"

namespace psk {

constexpr int NBK = 16;        // K-tile depth of the Newton products
constexpr int MAX_PROD = 8;    // products per Newton step (p <= 64)
constexpr int NTEMP = 5;

enum Phase { PH_INIT = 0, PH_ACTIVE = 1, PH_DONE = 2 };
enum BufId { ID_MI = 0, ID_MCUR, ID_MNEXT, ID_HCUR, ID_HNEXT, ID_T0 };
enum Epi { EPI_PLAIN = 0, EPI_NEWM = 1 };

struct NewtonBlock {
  const float* a;
  float* out;
  float* M[2];
  float* H[2];
  float* Mi;
  float* T[NTEMP];
  float* sumsq_partial;  // [npad/128 * npad/128] partial sums of ||D||_F^2
  int n;        // effective size
  int n_full;   // rows/cols of a and out
  int lda, ldo, npad, p;
  float alpha, one_minus_alpha, inv_p;
  // loop state
  int phase, cur, it, tries, total_iters, result_sel;
  float err, ratio, max_ev, ridge, ridge_try;
  unsigned err_bits;
  int power_iters;
  int symmetric;  // products are computed on the upper tile triangle and mirrored
};

struct NewtonTask {
  int block;
  short a_id, b_id, c_id, epi;
};

struct TileEntry {
  int task;
  short tm, tn;
};

struct HostStatus {
  int gen;
  int not_done;
  int need_init;
  int pad;
};

__device__ inline float* resolve(NewtonBlock* nb, int id) {
  switch (id) {
    case ID_MI: return nb->Mi;
    case ID_MCUR: return nb->M[nb->cur];
    case ID_MNEXT: return nb->M[nb->cur ^ 1];
    case ID_HCUR: return nb->H[nb->cur];
    case ID_HNEXT: return nb->H[nb->cur ^ 1];
    default: return nb->T[id - ID_T0];
  }
}

// ---- products -----------------------------------------------------------------
template <int BK>
__global__ __launch_bounds__(256, 2) void newton_stage_kernel(
    NewtonBlock* blocks, const NewtonTask* tasks, const TileEntry* tiles,
    int ntiles) {
  extern __shared__ __align__(16) float smem[];  // SmemCfg<BK>::TOTAL floats
  const TileEntry te = tiles[xcd_remap(blockIdx.x, ntiles)];
  const NewtonTask tk = tasks[te.task];
  NewtonBlock* nb = &blocks[tk.block];
  if (nb->phase != PH_ACTIVE) return;
  const int n = nb->n, ld = nb->npad;
  Operand A{resolve(nb, tk.a_id), ld, te.tm * TILE, ld, ld, true};
  Operand B{resolve(nb, tk.b_id), ld, te.tn * TILE, ld, ld, true};
  float* C = resolve(nb, tk.c_id);
  f32x16 acc[2][2];
  gemm_tile<KC, MC, BK, false>(A, B, n, smem, acc);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  if (tk.epi == EPI_PLAIN) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = te.tm * TILE + acc_row(wm, tm, r, lane);
          const int col = te.tn * TILE + acc_col(wn, tn, lane);
          gstore1(C + (int64_t)row * ld + col, acc[tm][tn][r]);
        }
    if (te.tm != te.tn && nb->symmetric)
      store_tile_transposed(acc, smem, C, nullptr, 0.f, ld, te.tn * TILE, te.tm * TILE);
    return;
  }
  // EPI_NEWM: C = new M; Mi = (1-alpha) I + alpha M (DS:844, two roundings as
  // written there); err = max |M - I| (DS:847) with the identity masked to n.
  float* Mi = nb->Mi;
  const float alpha = nb->alpha, oma = nb->one_minus_alpha;
  unsigned emax = 0;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = te.tm * TILE + acc_row(wm, tm, r, lane);
        const int col = te.tn * TILE + acc_col(wn, tn, lane);
        const float v = acc[tm][tn][r];
        const float ident = (row == col && row < n) ? 1.f : 0.f;
        gstore1(C + (int64_t)row * ld + col, v);
        gstore1(Mi + (int64_t)row * ld + col,
                __fadd_rn(__fmul_rn(oma, ident), __fmul_rn(alpha, v)));
        const unsigned e = abs_bits(__fsub_rn(v, ident));
        emax = e > emax ? e : emax;
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
    atomicMax(&nb->err_bits, m);
  }
  if (te.tm != te.tn && nb->symmetric) {
    __syncthreads();  // red[] lives in smem
    // off-diagonal tile: identity is 0 there, so Mi = fl(alpha * M)
    store_tile_transposed(acc, smem, C, Mi, alpha, ld, te.tn * TILE, te.tm * TILE);
  }
}

// ---- (re)initialisation of a try (DS:866-875) -----------------------------------
// pass 1: per 128x128 tile, partial sum of squares of D = A + ridge_try * I.
__global__ __launch_bounds__(256) void newton_init1_kernel(NewtonBlock* blocks,
                                                           const TileEntry* tiles) {
  __shared__ float red[4];
  const TileEntry te = tiles[blockIdx.x];
  NewtonBlock* nb = &blocks[te.task];
  if (nb->phase != PH_INIT) return;
  const int n = nb->n, tid = threadIdx.x;
  const float rt = nb->ridge_try;
  float ss = 0.f;
  const int tpr = nb->npad / TILE;
  for (int e = tid; e < TILE * TILE; e += 256) {
    const int row = te.tm * TILE + e / TILE, col = te.tn * TILE + e % TILE;
    if (row < n && col < n) {
      float d = nb->a[(int64_t)row * nb->lda + col];
      if (row == col) d = __fadd_rn(d, rt);
      ss += d * d;
    }
  }
  ss = wave_sum_f32(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  if (tid == 0)
    nb->sumsq_partial[te.tm * tpr + te.tn] = ((red[0] + red[1]) + red[2]) + red[3];
}

// pass 2: z, M0 = D z, H0 = I z^(1/p), Mi0, err0 = max|M0 - I|.
__global__ __launch_bounds__(256) void newton_init2_kernel(NewtonBlock* blocks,
                                                           const TileEntry* tiles) {
  __shared__ float bc;
  __shared__ unsigned red[4];
  const TileEntry te = tiles[blockIdx.x];
  NewtonBlock* nb = &blocks[te.task];
  if (nb->phase != PH_INIT) return;
  const int n = nb->n, ld = nb->npad, tid = threadIdx.x;
  const int tpr = ld / TILE;
  const float sumsq = fixed_order_sum_wave0(nb->sumsq_partial, tpr * tpr, tid, &bc);
  const float z = __fdiv_rn((float)(1 + nb->p), __fmul_rn(2.f, sqrtf(sumsq)));  // DS:870
  const float h0 = powf(z, nb->inv_p);                                          // DS:873
  const float rt = nb->ridge_try, alpha = nb->alpha, oma = nb->one_minus_alpha;
  float* M = nb->M[nb->cur];
  float* H0 = nb->H[nb->cur];
  float* H1 = nb->H[nb->cur ^ 1];
  float* Mi = nb->Mi;
  unsigned emax = 0;
  for (int e = tid; e < TILE * TILE; e += 256) {
    const int row = te.tm * TILE + e / TILE, col = te.tn * TILE + e % TILE;
    float m = 0.f, mi = 0.f, h = 0.f;
    if (row < n && col < n) {
      float d = nb->a[(int64_t)row * nb->lda + col];
      const float ident = row == col ? 1.f : 0.f;
      if (row == col) d = __fadd_rn(d, rt);                 // DS:869
      m = __fmul_rn(d, z);                                   // DS:871
      mi = __fadd_rn(__fmul_rn(oma, ident), __fmul_rn(alpha, m));
      h = __fmul_rn(ident, h0);
      const unsigned eb = abs_bits(__fsub_rn(m, ident));     // DS:872
      emax = eb > emax ? eb : emax;
    }
    const int64_t o = (int64_t)row * ld + col;
    M[o] = m;
    Mi[o] = mi;
    H0[o] = h;
    H1[o] = h;
  }
  emax = wave_max_u32(emax);
  if ((tid & 63) == 0) red[tid >> 6] = emax;
  __syncthreads();
  if (tid == 0) {
    unsigned m = red[0];
    m = red[1] > m ? red[1] : m;
    m = red[2] > m ? red[2] : m;
    m = red[3] > m ? red[3] : m;
    atomicMax(&nb->err_bits, m);
  }
}

// ---- loop control (single workgroup, one thread per block, grid-stride) --------
__device__ inline void finish_try(NewtonBlock* nb, float ridge_eps_unused) {
  // DS:878-882: error, is_converged select, retry decision.
  const bool conv = nb->ratio < 1.2f;
  nb->result_sel = conv ? nb->cur : (nb->cur ^ 1);
  nb->tries += 1;
  if (nb->err > 0.05f && nb->tries < 6) {
    nb->phase = PH_INIT;
    const float pow10[6] = {1.f, 10.f, 100.f, 1000.f, 10000.f, 100000.f};
    nb->ridge_try = __fmul_rn(nb->ridge, pow10[nb->tries]);  // DS:869
  } else {
    nb->phase = PH_DONE;
  }
}

// mode 0: after init2 (enter the inner loop, DS:874-877); mode 1: after one
// Newton step (DS:848 carry + DS:836-840 condition).
__global__ __launch_bounds__(256) void newton_control_kernel(
    NewtonBlock* blocks, int nblocks, int mode, int num_iters, float tol, int gen,
    HostStatus* status) {
  __shared__ int s_nd, s_ni;
  if (threadIdx.x == 0) { s_nd = 0; s_ni = 0; }
  __syncthreads();
  int nd = 0, ni = 0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    NewtonBlock* nb = &blocks[b];
    if (mode == 0 && nb->phase == PH_INIT) {
      nb->err = __uint_as_float(nb->err_bits);
      nb->err_bits = 0;
      nb->ratio = 1.f;
      nb->it = 0;
      const bool cont = nb->it < num_iters && nb->err > tol && nb->ratio < 1.2f;
      if (cont) nb->phase = PH_ACTIVE; else finish_try(nb, 0.f);
    } else if (mode == 1 && nb->phase == PH_ACTIVE) {
      const float new_err = __uint_as_float(nb->err_bits);
      nb->err_bits = 0;
      nb->ratio = __fdiv_rn(new_err, nb->err);
      nb->err = new_err;
      nb->it += 1;
      nb->total_iters += 1;
      nb->cur ^= 1;
      const bool cont = nb->it < num_iters && nb->err > tol && nb->ratio < 1.2f;
      if (!cont) finish_try(nb, 0.f);
    }
    nd += nb->phase != PH_DONE;
    ni += nb->phase == PH_INIT;
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
__global__ void newton_setup_kernel(NewtonBlock* blocks, const PiBlock* pis,
                                    int nblocks, float ridge_epsilon, int relative,
                                    const float* max_ev_given) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  NewtonBlock* nb = &blocks[b];
  float max_ev = 1.f;
  int pit = 0;
  if (relative) {
    if (max_ev_given) max_ev = max_ev_given[b];  // DS:815-817: lobpcg already has it
    else { max_ev = pis[b].lambda; pit = pis[b].iters; }
  }
  nb->max_ev = max_ev;
  nb->power_iters = pit;
  nb->ridge = __fmul_rn(ridge_epsilon, fmaxf(max_ev, 1e-25f));
  // fmaxf drops a NaN max_ev; jnp.maximum propagates it.
  if (max_ev != max_ev) nb->ridge = max_ev;
  nb->ridge_try = nb->ridge;  // * 10^0
}

// ---- result copy-out + metrics table (DS:902-907, 930-939) -----------------------
__global__ __launch_bounds__(256) void newton_final_kernel(NewtonBlock* blocks,
                                                           float* metrics) {
  NewtonBlock* nb = &blocks[blockIdx.x];
  const int n = nb->n, nf = nb->n_full;
  const int tid = blockIdx.y * 256 + threadIdx.x, nth = gridDim.y * 256;
  float* out = nb->out;
  if (n == 0) {
    for (int64_t e = tid; e < (int64_t)nf * nf; e += nth) {
      const int row = e / nf, col = e % nf;
      out[(int64_t)row * nb->ldo + col] = 0.f;
    }
  } else {
    const float* H = nb->H[nb->result_sel];
    for (int64_t e = tid; e < (int64_t)nf * nf; e += nth) {
      const int row = e / nf, col = e % nf;
      out[(int64_t)row * nb->ldo + col] =
          (row < n && col < n) ? H[(int64_t)row * nb->npad + col] : 0.f;
    }
  }
  if (tid == 0) {
    float* m = metrics + (int64_t)blockIdx.x * PS_METRICS_STRIDE;
    if (n == 0) {  // all padding: DS:930-937 (error forced to 0)
      m[PS_M_ERROR] = 0.f; m[PS_M_ITERS] = 0.f; m[PS_M_ERROR_RATIO] = 1.f;
      m[PS_M_RETRIES] = 1.f; m[PS_M_TOTAL_ITERS] = 0.f;
    } else {
      m[PS_M_ERROR] = nb->err; m[PS_M_ITERS] = (float)nb->it;
      m[PS_M_ERROR_RATIO] = nb->ratio; m[PS_M_RETRIES] = (float)nb->tries;
      m[PS_M_TOTAL_ITERS] = (float)nb->total_iters;
    }
    m[PS_M_MAX_EV] = nb->max_ev;
    m[PS_M_POWER_ITERS] = (float)nb->power_iters;
    m[PS_M_RESERVED] = 0.f;
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
  int nstages = 0;
  std::vector<std::vector<NewtonTask>> stage_tasks;
  std::vector<std::vector<TileEntry>> stage_tiles;
  std::vector<TileEntry> init_tiles;  // one per (block, tile)
  PiPlan pip;
  int max_n = 0;
  bool ok = true;
};

// All iterates of the coupled Newton iteration are polynomials in the (symmetric)
// input, hence symmetric and commuting: in symmetric mode only the tiles with
// tm <= tn of every product are computed and the strict upper ones are mirrored,
// which removes (T-1)/(2T) of the MFMA work (T = tiles per side).  The result
// differs from the full products of DS:845-846 only by which of the two rounded
// values x_ij / x_ji is kept.  PS_NEWTON_SYMMETRIC=0 restores the full products.
bool symmetric_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("PS_NEWTON_SYMMETRIC");
    mode = e ? (atoi(e) != 0) : 1;
  }
  return mode != 0;
}

void make_plan(Plan& pl, int batch, const int32_t* n, const int32_t* p,
               const int32_t* padding_start) {
  pl.batch = batch;
  pl.n_eff.resize(batch);
  pl.npad.resize(batch);
  pl.chains.resize(batch);
  for (int b = 0; b < batch; ++b) {
    int ne = n[b];
    if (padding_start) ne = std::max(0, std::min(ne, (int)padding_start[b]));
    pl.n_eff[b] = ne;
    pl.npad[b] = ne >= 1 ? psh::round_up(ne, TILE) : 0;
    pl.max_n = std::max(pl.max_n, ne);
    if (ne >= 1) {
      std::vector<Product> ch;
      int power_id;
      if (p[b] < 1 || !build_chain(p[b], ch, power_id)) { pl.ok = false; return; }
      // H <- H Mi rides in the first stage; M <- power @ M is last.
      std::vector<Product> full;
      full.push_back({ID_HCUR, ID_MI, ID_HNEXT});
      for (auto& q : ch) full.push_back(q);
      full.push_back({power_id, ID_MCUR, ID_MNEXT});
      pl.chains[b] = full;
    }
  }
  pl.pip.build(batch, pl.n_eff);
  // Stage s runs product s+1 of every block (product 0, the H update, joins
  // stage 0); the M update of a block runs as soon as its chain is done.
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
  pl.stage_tasks.assign(pl.nstages, {});
  pl.stage_tiles.assign(pl.nstages, {});
  for (int b = 0; b < batch; ++b) {
    auto& c = pl.chains[b];
    if (c.empty()) continue;
    const int t = pl.npad[b] / TILE;
    for (int tm = 0; tm < t; ++tm)
      for (int tn = 0; tn < t; ++tn) pl.init_tiles.push_back({b, (short)tm, (short)tn});
    for (size_t k = 0; k < c.size(); ++k) {
      const int s = stage_of(k, c.size());
      const bool last = k + 1 == c.size();
      NewtonTask tk{b, (short)c[k].a, (short)c[k].b, (short)c[k].c,
                    (short)(last ? EPI_NEWM : EPI_PLAIN)};
      const int tid = (int)pl.stage_tasks[s].size();
      pl.stage_tasks[s].push_back(tk);
      for (int tm = 0; tm < t; ++tm)
        for (int tn = (symmetric_mode() ? tm : 0); tn < t; ++tn)
          pl.stage_tiles[s].push_back({tid, (short)tm, (short)tn});
    }
  }
}

struct WsLayout {
  NewtonBlock* blocks;
  NewtonTask* tasks[MAX_PROD];
  TileEntry* tiles[MAX_PROD];
  TileEntry* init_tiles;
  std::vector<float*> mat[10];
  std::vector<float*> sumsq;
};

size_t carve(Plan& pl, Arena& ar, WsLayout* lo) {
  const int B = pl.batch;
  NewtonBlock* blocks = ar.take<NewtonBlock>(B);
  pl.pip.carve(ar, lo != nullptr);
  if (lo) { lo->blocks = blocks; }
  for (int s = 0; s < pl.nstages; ++s) {
    NewtonTask* t = ar.take<NewtonTask>(pl.stage_tasks[s].size());
    TileEntry* e = ar.take<TileEntry>(pl.stage_tiles[s].size());
    if (lo) { lo->tasks[s] = t; lo->tiles[s] = e; }
  }
  TileEntry* it = ar.take<TileEntry>(pl.init_tiles.size());
  if (lo) { lo->init_tiles = it; }
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

// Event pairs recorded on the caller's stream; resolved after the call.
struct ProfRun {
  struct Span { hipEvent_t a, b; int kind; };  // kind 0 stage, 1 power iter, 2 other
  std::vector<Span> spans;
  bool active;
  hipStream_t st;
  explicit ProfRun(hipStream_t s) : active(g_prof.on), st(s) {}
  void begin(int kind) {
    if (!active) return;
    Span sp; sp.kind = kind;
    if (hipEventCreate(&sp.a) != hipSuccess || hipEventCreate(&sp.b) != hipSuccess) { active = false; return; }
    (void)hipEventRecord(sp.a, st);
    spans.push_back(sp);
  }
  void end() {
    if (!active || spans.empty()) return;
    (void)hipEventRecord(spans.back().b, st);
  }
  void finish() {
    if (spans.empty()) return;
    (void)hipStreamSynchronize(st);
    for (auto& sp : spans) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
        if (sp.kind == 0) { g_prof.stage_ms += ms; g_prof.stage_launches += 1; }
        else if (sp.kind == 1) g_prof.pi_ms += ms;
        else g_prof.other_ms += ms;
      }
      (void)hipEventDestroy(sp.a);
      (void)hipEventDestroy(sp.b);
    }
    spans.clear();
  }
};

// One mapped status ring per host thread: a thread runs one call at a time, so calls
// on distinct (stream, workspace) pairs from different threads never share slots.
HostStatus* pinned_status() {
  static thread_local HostStatus* st = nullptr;
  if (!st) {
    if (hipHostMalloc((void**)&st, 64 * sizeof(HostStatus), hipHostMallocMapped) !=
        hipSuccess)
      st = nullptr;
  }
  return st;
}

bool vec_ok(const float* p, int ld) {
  return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0);
}

}  // namespace

extern "C" size_t ps_newton_root_workspace_bytes(int batch, const int32_t* n,
                                                 const int32_t* p,
                                                 const int32_t* padding_start) {
  if (batch <= 0 || !n || !p) return 0;
  Plan pl;
  make_plan(pl, batch, n, p, padding_start);
  if (!pl.ok) return 0;
  Arena ar(nullptr, 0);
  return carve(pl, ar, nullptr) + 256;
}

static int newton_driver(
    void* stream, const float* const* a, const int32_t* n, const int32_t* lda,
    const int32_t* p, const int32_t* padding_start, int batch, int num_iters,
    float ridge_epsilon, float error_tolerance, int relative_matrix_epsilon,
    const float* max_ev_given, float* const* out, const int32_t* ldo, float* metrics,
    void* workspace, size_t workspace_bytes, int32_t* iters_executed_host) {
  PS_DEVICE_CHECK();
  if (batch <= 0 || !a || !n || !lda || !p || !out || !ldo || !metrics || !workspace ||
      num_iters < 1)
    return PS_EINVAL;
  for (int b = 0; b < batch; ++b)
    if (n[b] < 1 || lda[b] < n[b] || ldo[b] < n[b] || !a[b] || !out[b]) return PS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  Plan pl;
  make_plan(pl, batch, n, p, padding_start);
  if (!pl.ok) return PS_EUNSUPPORTED;
  if (pl.max_n > 16384) return PS_EUNSUPPORTED;
  Arena ar(workspace, workspace_bytes);
  WsLayout lo;
  carve(pl, ar, &lo);
  if (ar.overflow) return PS_EWORKSPACE;
  HostStatus* status = pinned_status();
  if (!status) return PS_EINTERNAL;

  // ---- upload plan ----------------------------------------------------------
  std::vector<NewtonBlock> hb(batch);
  for (int b = 0; b < batch; ++b) {
    NewtonBlock& nb = hb[b];
    memset(&nb, 0, sizeof(nb));
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
    nb.phase = pl.n_eff[b] >= 1 ? PH_INIT : PH_DONE;
    nb.ratio = 1.f;
    nb.symmetric = symmetric_mode() ? 1 : 0;
  }

  PS_HIP(hipMemcpyAsync(lo.blocks, hb.data(), sizeof(NewtonBlock) * batch,
                        hipMemcpyHostToDevice, st));
  for (int s = 0; s < pl.nstages; ++s) {
    PS_HIP(hipMemcpyAsync(lo.tasks[s], pl.stage_tasks[s].data(),
                          sizeof(NewtonTask) * pl.stage_tasks[s].size(),
                          hipMemcpyHostToDevice, st));
    PS_HIP(hipMemcpyAsync(lo.tiles[s], pl.stage_tiles[s].data(),
                          sizeof(TileEntry) * pl.stage_tiles[s].size(),
                          hipMemcpyHostToDevice, st));
  }
  if (!pl.init_tiles.empty())
    PS_HIP(hipMemcpyAsync(lo.init_tiles, pl.init_tiles.data(),
                          sizeof(TileEntry) * pl.init_tiles.size(),
                          hipMemcpyHostToDevice, st));
  // The host vectors above must outlive the async copies (pageable memory is
  // staged synchronously by the runtime, but do not rely on it).
  PS_HIP(hipStreamSynchronize(st));
  {
    int rc = pl.pip.upload(st, a, lda);
    if (rc) return rc;
  }

  ProfRun prof(st);
  // ---- power iteration -> ridge epsilon --------------------------------------
  prof.begin(1);
  if (relative_matrix_epsilon && !max_ev_given) {
    int rc = pl.pip.enqueue(st, 100, 1e-6f);  // DS:820-825
    if (rc) return rc;
  }
  prof.end();
  hipLaunchKernelGGL(newton_setup_kernel, dim3((batch + 255) / 256), dim3(256), 0, st,
                     lo.blocks, pl.pip.d_blocks, batch, ridge_epsilon,
                     relative_matrix_epsilon, max_ev_given);
  PS_LAUNCH_CHECK();

  // ---- Newton loop -------------------------------------------------------------
  // K-tile depth of the product kernel: 32 (73.7 KB LDS => exactly 2 workgroups per
  // CU, half the barriers) or 16 (40 KB, 3 per CU).  PS_NEWTON_BK overrides.
  static int stage_bk = 0;
  if (stage_bk == 0) {
    const char* e = getenv("PS_NEWTON_BK");
    stage_bk = (e && atoi(e) == 16) ? 16 : ((e && atoi(e) == 32) ? 32 : NBK);
    if (hipFuncSetAttribute((const void*)newton_stage_kernel<32>,
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(SmemCfg<32>::TOTAL * sizeof(float))) != hipSuccess)
      stage_bk = 16;
  }
  const int ninit = (int)pl.init_tiles.size();
  int executed = 0;
  if (ninit > 0) {
    hipEvent_t ev[2];
    PS_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    {
      const hipError_t e1 = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
      if (e1 != hipSuccess) { (void)hipEventDestroy(ev[0]); return (int)e1; }
    }
    const int cap = 6 * (num_iters + 2) + 4;
    bool need_init = true;
    int rc = 0;
    for (int g = 0; g < cap; ++g) {
      HostStatus* slot = &status[g % 64];
      slot->gen = -1;
      if (need_init) {
        prof.begin(2);
        hipLaunchKernelGGL(newton_init1_kernel, dim3(ninit), dim3(256), 0, st, lo.blocks,
                           lo.init_tiles);
        hipLaunchKernelGGL(newton_init2_kernel, dim3(ninit), dim3(256), 0, st, lo.blocks,
                           lo.init_tiles);
        hipLaunchKernelGGL(newton_control_kernel, dim3(1), dim3(256), 0, st, lo.blocks,
                           batch, 0, num_iters, error_tolerance, g, (HostStatus*)nullptr);
        prof.end();
      }
      for (int s = 0; s < pl.nstages; ++s) {
        const int nt = (int)pl.stage_tiles[s].size();
        prof.begin(0);
        if (stage_bk == 32)
          hipLaunchKernelGGL(newton_stage_kernel<32>, dim3(nt), dim3(256),
                             SmemCfg<32>::TOTAL * sizeof(float), st, lo.blocks, lo.tasks[s],
                             lo.tiles[s], nt);
        else
          hipLaunchKernelGGL(newton_stage_kernel<16>, dim3(nt), dim3(256),
                             SmemCfg<16>::TOTAL * sizeof(float), st, lo.blocks, lo.tasks[s],
                             lo.tiles[s], nt);
        prof.end();
      }
      hipLaunchKernelGGL(newton_control_kernel, dim3(1), dim3(256), 0, st, lo.blocks,
                         batch, 1, num_iters, error_tolerance, g, slot);
      if ((rc = (int)hipGetLastError()) != 0) break;
      if ((rc = (int)hipEventRecord(ev[g & 1], st)) != 0) break;
      ++executed;
      if (g >= 1) {
        if ((rc = (int)hipEventSynchronize(ev[(g - 1) & 1])) != 0) break;
        const HostStatus seen = status[(g - 1) % 64];
        if (seen.gen != g - 1) { rc = PS_EINTERNAL; break; }
        if (seen.not_done == 0) break;
        need_init = seen.need_init > 0;
      } else {
        need_init = false;
      }
    }
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    if (rc) return rc;
  }
  prof.begin(2);
  hipLaunchKernelGGL(newton_final_kernel, dim3(batch, 32), dim3(256), 0, st, lo.blocks,
                     metrics);
  prof.end();
  PS_LAUNCH_CHECK();
  prof.finish();
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
    float* out_lambda, int32_t* out_iters, float* out_v, int32_t ldv, void* workspace,
    size_t workspace_bytes) {
  PS_DEVICE_CHECK();
  if (batch <= 0 || !a || !n || !lda || !out_lambda || !workspace || num_iters < 1)
    return PS_EINVAL;
  for (int b = 0; b < batch; ++b)
    if (n[b] < 1 || lda[b] < n[b] || !a[b]) return PS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  PiPlan pp;
  pi_plan_from_args(pp, batch, n, padding_start);
  if (pp.max_n > 16384) return PS_EUNSUPPORTED;
  if (out_v && ldv < pp.max_n) return PS_EINVAL;
  Arena ar(workspace, workspace_bytes);
  pp.carve(ar, true);
  if (ar.overflow) return PS_EWORKSPACE;
  int rc = pp.upload(st, a, lda);
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
    float ridge_epsilon, float error_tolerance, int relative_matrix_epsilon,
    float* const* out, const int32_t* ldo, float* metrics, void* workspace,
    size_t workspace_bytes, int32_t* iters_executed_host) {
  return newton_driver(stream, a, n, lda, p, padding_start, batch, num_iters, ridge_epsilon,
                       error_tolerance, relative_matrix_epsilon, nullptr, out, ldo, metrics,
                       workspace, workspace_bytes, iters_executed_host);
}

extern "C" int ps_newton_root_batched_maxev_f32(
    void* stream, const float* const* a, const int32_t* n, const int32_t* lda,
    const int32_t* p, const int32_t* padding_start, int batch, int num_iters,
    float ridge_epsilon, float error_tolerance, const float* max_ev, float* const* out,
    const int32_t* ldo, float* metrics, void* workspace, size_t workspace_bytes,
    int32_t* iters_executed_host) {
  if (!max_ev) return PS_EINVAL;
  return newton_driver(stream, a, n, lda, p, padding_start, batch, num_iters, ridge_epsilon,
                       error_tolerance, 1, max_ev, out, ldo, metrics, workspace,
                       workspace_bytes, iters_executed_host);
}
