// eigh_cj.hip.h — round-3 eigensolver of the eigh root path (included by eigh.hip inside
// namespace psk): one-sided (Hestenes) block Jacobi on the Cholesky factor.
//
//   D = A + ridge I (DS:1005-1006)  =  L L^T      float64-accumulated blocked Cholesky, L stored
//                                                 as float32 in G
//   sweeps over round-robin pairs (I, J) of 64-wide block columns of G:
//       P = [G_I G_J]^T [G_I G_J]        128 x 128 Gram matrix        (cj_gram_kernel, MFMA f32)
//       Q = eigenvectors of P            one-sided Jacobi in LDS      (cj_pivot_kernel)
//       [G_I G_J] <- [G_I G_J] Q         128-row tiles, K = 128       (cj_update_kernel, MFMA f32)
//   until every scaled Gram entry |g_i . g_j| / (|g_i| |g_j|) seen in a sweep is below `tol`.
//   Then D = G G^T = U diag(|g_j|^2) U^T with U = the normalised columns of G: the eigenvectors
//   need no accumulation and there is no left-side update, 6 n^3 flops per sweep instead of the
//   12 n^3 of the two-sided method (A <- J^T A, A <- A J, V <- V J), and the rotations act on a
//   factor whose condition number is the square root of D's (Demmel-Veselic: the small
//   eigenvalues of a graded / rank-deficient-plus-ridge statistic keep their relative accuracy).
//   Prototype with sweep counts and accuracy: tools/proto_onesided_chol.py.
//
// Replaces jnp.linalg.eigh at DS:1007 (LAPACK ssyevd on the reference's CPU path) for blocks
// of more than 128 rows in root mode; the blocked two-sided solver stays as the fallback for a
// block whose Cholesky factorisation breaks down (an input that is not positive definite after
// the ridge) and for plain eigenpairs of possibly indefinite matrices (ps_eigh_batched_f32).
//
// Workspace aliases (no extra memory): G = eb->X, the Gram matrices P of a round = eb->V
// (unused until the eigenvectors are written there at the end), the float64 Cholesky panel and
// the inverse of its diagonal block = eb->W, the per-(round, pair) skip flags = eb->offpart.
#pragma once

constexpr int CB = 64;   // Cholesky tile

// ---- Cholesky, step j: S_ij = D_ij - sum_{k < j} L_ik L_jk^T for the tiles i >= j of the
// panel, accumulated in float64 on v_mfma_f64_16x16x4_f64 (operands are float32: their products
// are exact in float64), written as float64 to the panel buffer Sp[npad][64].
__global__ __launch_bounds__(256) void cj_chol_schur_kernel(EighBlock* blocks, const ETile* tiles,
                                                            int j) {
  __shared__ float sL[RK][RQ + 1];
  __shared__ float sR[RK][RQ + 1];
  const ETile te = tiles[blockIdx.x];
  EighBlock* eb = &blocks[te.block];
  const int i = te.k;
  if (!eb->cj_active || i < j || j >= eb->npad / CB) return;
  const int ld = eb->npad, tid = threadIdx.x;
  const int i0 = i * CB, j0 = j * CB;
  const float* G = eb->X;
  const int wave = tid >> 6, lane = tid & 63;
  const int qr = 32 * (wave >> 1), qc = 32 * (wave & 1);
  const int fi = lane & 15, fk = lane >> 4;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < j0; k0 += RK) {
    const int m = tid >> 2, k4 = (tid & 3) * 4;
    const f32x4 v = gload4(G + (int64_t)(i0 + m) * ld + k0 + k4);
    const f32x4 w = gload4(G + (int64_t)(j0 + m) * ld + k0 + k4);
    sL[k4 + 0][m] = v[0]; sL[k4 + 1][m] = v[1]; sL[k4 + 2][m] = v[2]; sL[k4 + 3][m] = v[3];
    sR[k4 + 0][m] = w[0]; sR[k4 + 1][m] = w[1]; sR[k4 + 2][m] = w[2]; sR[k4 + 3][m] = w[3];
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < RK; kk += 4) {
      double af[2], bf[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = (double)sL[kk + fk][qr + 16 * a + fi];
#pragma unroll
      for (int b = 0; b < 2; ++b) bf[b] = (double)sR[kk + fk][qc + 16 * b + fi];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }
  double* Sp = reinterpret_cast<double*>(eb->W);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int r = qr + 16 * a + fk + 4 * v, c = qc + 16 * b + fi;
        const double d = (double)gload1(eb->D + (int64_t)(i0 + r) * ld + j0 + c);
        Sp[(int64_t)(i0 + r) * CB + c] = d - acc[a][b][v];
      }
}

// ---- Cholesky, step j: L_jj = chol(S_jj) in float64 in LDS, rounded to float32 into G (upper
// part zero); the inverse of the ROUNDED factor (float64) goes to the scratch behind the panel.
// Rows beyond the block's true size are an identity for the arithmetic and zero in G.  A
// non-positive or non-finite pivot marks the block (chol_fail): it is solved by the two-sided
// fallback.  One workgroup per block.
__global__ __launch_bounds__(256) void cj_chol_potrf_kernel(EighBlock* blocks, const int* ids,
                                                            int j) {
  __shared__ double S[CB][CB + 1];
  __shared__ double Xi[CB][CB + 1];
  __shared__ double s_d;
  EighBlock* eb = &blocks[ids[blockIdx.x]];
  if (!eb->cj_active || j >= eb->npad / CB) return;
  const int ld = eb->npad, n = eb->n, tid = threadIdx.x, j0 = j * CB;
  double* Sp = reinterpret_cast<double*>(eb->W);
  for (int e = tid; e < CB * CB; e += 256) {
    const int r = e >> 6, c = e & 63;
    double v = Sp[(int64_t)(j0 + r) * CB + c];
    if (j0 + r >= n || j0 + c >= n) v = r == c ? 1.0 : 0.0;
    S[r][c] = v;
  }
  __syncthreads();
  bool fail = false;
  for (int c = 0; c < CB; ++c) {
    if (tid == 0) {
      const double d = S[c][c];
      const bool ok = d > 0.0 && d < 1.0e300;
      s_d = ok ? sqrt(d) : 1.0;
      if (!ok) eb->chol_fail = 1;
    }
    __syncthreads();
    const double d = s_d;
    if (tid > c && tid < CB) S[tid][c] = S[tid][c] / d;
    if (tid == 0) S[c][c] = d;
    __syncthreads();
    const int w = CB - 1 - c;
    for (int e = tid; e < w * w; e += 256) {
      const int r = c + 1 + e / w, cc = c + 1 + e % w;
      if (cc <= r) S[r][cc] = fma(-S[r][c], S[cc][c], S[r][cc]);
    }
    __syncthreads();
  }
  (void)fail;
  float* G = eb->X;
  for (int e = tid; e < CB * CB; e += 256) {
    const int r = e >> 6, c = e & 63;
    float v = 0.f;
    if (c <= r && j0 + r < n) v = (float)S[r][c];
    gstore1(G + (int64_t)(j0 + r) * ld + j0 + c, v);
    if (c <= r) S[r][c] = (j0 + r < n) ? (double)v : (r == c ? 1.0 : 0.0);   // the rounded factor
    else S[r][c] = 0.0;
  }
  __syncthreads();
  // X = L^-1 (lower triangular), one column per thread
  if (tid < CB) {
    const int c = tid;
    for (int r = 0; r < c; ++r) Xi[r][c] = 0.0;
    Xi[c][c] = 1.0 / S[c][c];
    for (int r = c + 1; r < CB; ++r) {
      double s = 0.0;
      for (int k = c; k < r; ++k) s = fma(S[r][k], Xi[k][c], s);
      Xi[r][c] = -s / S[r][r];
    }
  }
  __syncthreads();
  double* Linv = Sp + (int64_t)ld * CB;
  for (int e = tid; e < CB * CB; e += 256) Linv[e] = Xi[e >> 6][e & 63];
}

// ---- Cholesky, step j: L_ij = S_ij L_jj^-T for the tiles i > j, float64, rounded into G.
__global__ __launch_bounds__(256) void cj_chol_trsm_kernel(EighBlock* blocks, const ETile* tiles,
                                                           int j) {
  __shared__ double S[CB][CB + 1];
  __shared__ double Xi[CB][CB + 1];
  const ETile te = tiles[blockIdx.x];
  EighBlock* eb = &blocks[te.block];
  const int i = te.k;
  if (!eb->cj_active || i <= j || j >= eb->npad / CB) return;
  const int ld = eb->npad, tid = threadIdx.x, i0 = i * CB, j0 = j * CB;
  const double* Sp = reinterpret_cast<const double*>(eb->W);
  const double* Linv = Sp + (int64_t)ld * CB;
  for (int e = tid; e < CB * CB; e += 256) {
    const int r = e >> 6, c = e & 63;
    S[r][c] = Sp[(int64_t)(i0 + r) * CB + c];
    Xi[r][c] = Linv[e];
  }
  __syncthreads();
  float* G = eb->X;
  const int m = tid >> 2;
#pragma unroll 4
  for (int q = 0; q < 16; ++q) {
    const int c = (tid & 3) + 4 * q;
    double s = 0.0;
    for (int k = 0; k <= c; ++k) s = fma(S[m][k], Xi[c][k], s);
    gstore1(G + (int64_t)(i0 + m) * ld + j0 + c, (float)s);
  }
}

// zero the part of G above the diagonal that the factorisation never writes (the workspace is
// not cleared by the caller): whole 128-tiles with k < t, and the upper-right 64 x 64 quadrant
// of the diagonal 128-tiles (cj_chol_potrf_kernel writes the 64 x 64 diagonal tiles in full).
__global__ __launch_bounds__(256) void cj_zero_upper_kernel(EighBlock* blocks, const ETile* tiles) {
  const ETile te = tiles[blockIdx.x];
  EighBlock* eb = &blocks[te.block];
  if (!eb->cj_active || te.k > te.t) return;
  const int ld = eb->npad, tid = threadIdx.x;
  float* G = eb->X;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const bool diag = te.k == te.t;
  for (int e = tid; e < TILE * TILE / 4; e += 256) {
    const int r = e >> 5, c4 = (e & 31) * 4;
    if (diag && !(r < CB && c4 >= CB)) continue;
    *(f32x4 PS_GLOBAL*)(G + (int64_t)(te.k * TILE + r) * ld + te.t * TILE + c4) = z;
  }
}

// which blocks take this path; the two-sided solver skips them (active = 0)
__global__ void cj_select_kernel(EighBlock* blocks, int nblocks) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  EighBlock* eb = &blocks[b];
  const int on = (eb->n > 0 && !eb->small && eb->npad >= 2 * TILE && !eb->td_done) ? 1 : 0;
  eb->cj = on;
  eb->cj_active = on;
  eb->chol_fail = 0;
  if (on) eb->active = 0;
}

// ---- P = [G_I G_J]^T [G_I G_J]: one workgroup per (block, pair); the K-tile of the panel
// (16 rows x the 128 gathered columns) is staged ONCE and serves as both MFMA operands.  P is
// symmetric: only the 10 upper 32 x 32 tiles (ti <= tj) are computed, dealt 3 + 3 + 2 + 2 to
// the four wavefronts (the critical wavefront issues 3/4 of the MFMAs of the full product), and
// only they are stored -- cj_pivot_kernel reads the upper triangle.
template <int W> struct GramTiles;
template <> struct GramTiles<0> { static constexpr int N = 3; static constexpr int TI[3] = {0, 0, 0}, TJ[3] = {0, 1, 2}; };
template <> struct GramTiles<1> { static constexpr int N = 3; static constexpr int TI[3] = {0, 1, 1}, TJ[3] = {3, 1, 2}; };
template <> struct GramTiles<2> { static constexpr int N = 2; static constexpr int TI[3] = {1, 2, 0}, TJ[3] = {3, 2, 0}; };
template <> struct GramTiles<3> { static constexpr int N = 2; static constexpr int TI[3] = {2, 3, 0}, TJ[3] = {3, 3, 0}; };

template <int W>
__device__ __forceinline__ void cj_gram_body(const float* G, int ld, const int (&goff)[2],
                                             const int (&soff)[2], float* smem, float* P,
                                             int lane) {
  using T = GramTiles<W>;
  constexpr int BK = 16, LD = SmemCfg<BK>::MC_LD;
  f32x16 acc[T::N];
#pragma unroll
  for (int t = 0; t < T::N; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int nk = ld / BK;
  const int i = lane & 31, h = lane >> 5;
  f32x4 r[2];
#pragma unroll
  for (int v = 0; v < 2; ++v) r[v] = gload4(G + goff[v]);
#pragma unroll
  for (int v = 0; v < 2; ++v) *reinterpret_cast<f32x4*>(smem + soff[v]) = r[v];
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const float* cur = smem + (kt & 1) * BK * LD;
    float* nxt = smem + ((kt + 1) & 1) * BK * LD;
    const bool more = kt + 1 < nk;
    if (more) {
#pragma unroll
      for (int v = 0; v < 2; ++v) r[v] = gload4(G + (int64_t)(kt + 1) * BK * ld + goff[v]);
    }
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      float f[4][4];   // fragment of column block b at k = 8c + 4h + s (unused blocks fold away)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) f[b][s2] = cur[(8 * c + 4 * h + s2) * LD + b * 32 + i];
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
        for (int t = 0; t < T::N; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[T::TI[t]][s2], f[T::TJ[t]][s2], acc[t],
                                                        0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int v = 0; v < 2; ++v) *reinterpret_cast<f32x4*>(nxt + soff[v]) = r[v];
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < T::N; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = T::TI[t] * 32 + (q & 3) + 8 * (q >> 2) + 4 * h, col = T::TJ[t] * 32 + i;
      gstore1(P + row * JP + col, acc[t][q]);
    }
}

__global__ __launch_bounds__(256, 2) void cj_gram_kernel(EighBlock* blocks, const ETile* tiles,
                                                         int ntiles, int round) {
  constexpr int BK = 16, LD = SmemCfg<BK>::MC_LD;   // [16][132]
  __shared__ __align__(16) float smem[2 * BK * LD];
  const ETile te = tiles[xcd_remap(blockIdx.x, ntiles)];
  EighBlock* eb = &blocks[te.block];
  if (!eb->cj_active || round >= eb->nb - 1) return;
  int I, J;
  rr_pair(eb->nb, round, te.k, I, J);
  const int ld = eb->npad, tid = threadIdx.x;
  const float* G = eb->X;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // thread -> two float4 of a K-tile: rows k = f >> 5, columns 4 * (f & 31) of the gathered panel
  int goff[2], soff[2];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int f = tid + 256 * v, k = f >> 5, c4 = (f & 31) * 4;
    const int gc = c4 < JB ? I * JB + c4 : J * JB + (c4 - JB);
    goff[v] = k * ld + gc;
    soff[v] = k * LD + c4;
  }
  float* P = eb->V + (int64_t)te.k * JP * JP;
  switch (wave) {
    case 0: cj_gram_body<0>(G, ld, goff, soff, smem, P, lane); break;
    case 1: cj_gram_body<1>(G, ld, goff, soff, smem, P, lane); break;
    case 2: cj_gram_body<2>(G, ld, goff, soff, smem, P, lane); break;
    default: cj_gram_body<3>(G, ld, goff, soff, smem, P, lane); break;
  }
}

// ---- the same Gram matrix on the bf16 MFMA for the sweeps that are far from convergence: the panel
// is split into bf16 hi / lo pairs (x = hi + lo) and P = hi^T hi + hi^T lo + lo^T hi accumulates in
// float32 -- ~2^-17 relative to |g_i||g_j| per entry, i.e. scaled entries good to ~1e-5, which is
// all a rotation needs while the largest scaled entry of the sweep is above 1e-3 (the pivot's Q is
// orthogonal whatever P it was computed from, so G G^T = D is untouched; the STOP decision only
// ever comes from a float32 sweep: the noise floor of this kernel is above the sweep tolerance, so
// a block can never converge on it).  3 / 16 of the float32 MFMA time: the kernel is HBM-bound (it
// streams the pair's panel once).  K-tile = 32 rows of the panel; every thread loads a 4 x 4
// patch (4 rows x 16 bytes), converts and writes it TRANSPOSED -- [column][k], k-contiguous,
// 4 x ds_write_b64 per plane -- so that an MFMA fragment (8 consecutive k of one column) is one
// 16-byte LDS read; rows of 40 bf16 (80 bytes) keep those reads conflict-free.
template <int W, int PLANES>
__device__ __forceinline__ void cj_gram_x3_mfma(const uint16_t* sh, const uint16_t* sl, const uint16_t* s2,
                                                int lane, f32x16 (&acc)[3]) {
  using T = GramTiles<W>;
  const int i = lane & 31, h8 = 8 * (lane >> 5);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    xbf16x8 fh[4], fl[4], f2[4];   // fragments of column block b (unused blocks fold away)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      fh[b] = *reinterpret_cast<const xbf16x8*>(sh + (b * 32 + i) * XLD + ks * 16 + h8);
      fl[b] = *reinterpret_cast<const xbf16x8*>(sl + (b * 32 + i) * XLD + ks * 16 + h8);
      if (PLANES == 3) f2[b] = *reinterpret_cast<const xbf16x8*>(s2 + (b * 32 + i) * XLD + ks * 16 + h8);
    }
#pragma unroll
    for (int t = 0; t < T::N; ++t) {
      f32x16 c = acc[t];
      if (PLANES == 3) {   // the three terms of weight 2^-16: x = h + l + l2, products above 2^-24 kept
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2[T::TI[t]], fh[T::TJ[t]], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[T::TI[t]], fl[T::TJ[t]], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[T::TI[t]], f2[T::TJ[t]], c, 0, 0, 0);
      }
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[T::TI[t]], fh[T::TJ[t]], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[T::TI[t]], fl[T::TJ[t]], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[T::TI[t]], fh[T::TJ[t]], c, 0, 0, 0);
      acc[t] = c;
    }
  }
}

// PLANES = 3: three bf16 planes per operand and six products -- float32-level accuracy (the
// arithmetic of cj_update_x6_kernel), still HBM-bound: the float32 MFMA form of the Gram kernel
// shares its FMA lanes with the pivots of the other stream group, this one does not.
template <int PLANES>
__global__ __launch_bounds__(256, 2) void cj_gram_x3_kernel(EighBlock* blocks, const ETile* tiles,
                                                            int ntiles, int round) {
  // two stages x PLANES planes x [128 columns][40] bf16 = 40,960 (61,440) bytes
  __shared__ __align__(16) uint16_t simg[2][PLANES][JP * XLD];
  const ETile te = tiles[xcd_remap(blockIdx.x, ntiles)];
  EighBlock* eb = &blocks[te.block];
  if (!eb->cj_active || round >= eb->nb - 1) return;
  int I, J;
  rr_pair(eb->nb, round, te.k, I, J);
  const int ld = eb->npad, tid = threadIdx.x;
  const float* G = eb->X;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // thread -> a 4-row x 4-column patch of a K-tile: rows 4 * (tid >> 5) .., columns 4 * (tid & 31) ..
  const int rg = tid >> 5, c4 = (tid & 31) * 4;
  const int gc = c4 < JB ? I * JB + c4 : J * JB + (c4 - JB);
  const float* gp = G + (int64_t)(4 * rg) * ld + gc;
  const int nk = ld / 32;
  f32x4 r[4];
  auto load = [&](int kt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = gload4(gp + (int64_t)(kt * 32 + q) * ld);
  };
  auto store = [&](int stage) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {   // column c4 + e: its 4 consecutive k as one 8-byte write per plane
      xbf16x4 hi, lo, l2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float x = r[q][e];
        const __bf16 hb = (__bf16)x;
        const __bf16 lb = (__bf16)(x - (float)hb);
        hi[q] = hb;
        lo[q] = lb;
        l2[q] = (__bf16)((x - (float)hb) - (float)lb);
      }
      *reinterpret_cast<xbf16x4*>(&simg[stage][0][(c4 + e) * XLD + 4 * rg]) = hi;
      *reinterpret_cast<xbf16x4*>(&simg[stage][1][(c4 + e) * XLD + 4 * rg]) = lo;
      if (PLANES == 3) *reinterpret_cast<xbf16x4*>(&simg[stage][2][(c4 + e) * XLD + 4 * rg]) = l2;
    }
  };
  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  load(0);
  store(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) load(kt + 1);
    switch (wave) {
      case 0: cj_gram_x3_mfma<0, PLANES>(simg[cur][0], simg[cur][1], simg[cur][PLANES - 1], lane, acc); break;
      case 1: cj_gram_x3_mfma<1, PLANES>(simg[cur][0], simg[cur][1], simg[cur][PLANES - 1], lane, acc); break;
      case 2: cj_gram_x3_mfma<2, PLANES>(simg[cur][0], simg[cur][1], simg[cur][PLANES - 1], lane, acc); break;
      default: cj_gram_x3_mfma<3, PLANES>(simg[cur][0], simg[cur][1], simg[cur][PLANES - 1], lane, acc); break;
    }
    if (more) store(cur ^ 1);
    __syncthreads();
  }
  float* P = eb->V + (int64_t)te.k * JP * JP;
  const int i = lane & 31, h = lane >> 5;
  auto put = [&](auto tag) {
    using T = decltype(tag);
#pragma unroll
    for (int t = 0; t < T::N; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = T::TI[t] * 32 + (q & 3) + 8 * (q >> 2) + 4 * h, col = T::TJ[t] * 32 + i;
        gstore1(P + row * JP + col, acc[t][q]);
      }
  };
  switch (wave) {
    case 0: put(GramTiles<0>{}); break;
    case 1: put(GramTiles<1>{}); break;
    case 2: put(GramTiles<2>{}); break;
    default: put(GramTiles<3>{}); break;
  }
}

// ---- all inner sweeps of the one-sided Jacobi on a 128-column problem with ONE column of every
// pair kept in registers.  A sweep is a recursive-halving tournament: for block sizes s = 128,
// 64, ..., 2 the first half of every block of s columns stays put ("stationary": its G and V
// columns live in the registers of its 16-lane group for the s/2 rounds of the level) and meets
// the columns of the second half one after the other (round r: column i meets s/2 + (i + r) mod
// s/2); 64 + 32 + ... + 1 = 127 rounds of 64 disjoint pairs, every pair once.  Only the moving
// column of a pair is read from and written back to LDS in a round: half the LDS traffic of
// onesided_jacobi_lds_t (whose round-robin tournament moves both columns), which is what bounds
// that kernel (ds_write_b128 ~79 B/clk).  Same rotation arithmetic.  1024 threads, m = 128.
__device__ inline int onesided_jacobi_lds_stationary(float* G, float* V, int* s_rot,
                                                     int max_sweeps, float done_cos2) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane >> 4, l = lane & 15;
  const int k = 4 * wave + sub;                 // pair index within a round, 0..63
  int sweeps_total = 0;
  for (int sweeps = 0; sweeps < max_sweeps; ++sweeps, ++sweeps_total) {
    float rotated = 0.f;
    for (int half = 64; half >= 1; half >>= 1) {   // block size s = 2 * half
      const int blk = k / half, i = k - blk * half;
      const int p = blk * 2 * half + i;
      float* gp = G + p * SE_LD + 4 * l;
      float* vp = V + p * SE_LD + 4 * l;
      f32x4 a0 = *reinterpret_cast<f32x4*>(gp), a1 = *reinterpret_cast<f32x4*>(gp + 64);
      f32x4 w0 = *reinterpret_cast<f32x4*>(vp), w1 = *reinterpret_cast<f32x4*>(vp + 64);
      float aa = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) aa += a0[j] * a0[j] + a1[j] * a1[j];
      aa = row16_sum(aa);
      for (int r = 0; r < half; ++r) {
        int iq = i + r; if (iq >= half) iq -= half;
        const int q = blk * 2 * half + half + iq;
        float* gq = G + q * SE_LD + 4 * l;
        float* vq = V + q * SE_LD + 4 * l;
        const f32x4 b0 = *reinterpret_cast<f32x4*>(gq), b1 = *reinterpret_cast<f32x4*>(gq + 64);
        const f32x4 x0 = *reinterpret_cast<f32x4*>(vq), x1 = *reinterpret_cast<f32x4*>(vq + 64);
        float bb = 0.f, ab = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bb += b0[j] * b0[j] + b1[j] * b1[j];
          ab += a0[j] * b0[j] + a1[j] * b1[j];
        }
        bb = row16_sum(bb); ab = row16_sum(ab);
        if (fabsf(ab) > 3e-7f * __builtin_amdgcn_sqrtf(aa * bb)) {
          const float zeta = (bb - aa) * __builtin_amdgcn_rcpf(2.f * ab);
          const float t = copysignf(1.f, zeta) *
                          __builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(1.f + zeta * zeta));
          const float c = __builtin_amdgcn_rsqf(1.f + t * t), sn = c * t;
          if (c == c && sn == sn) {
            f32x4 nb0, nb1, nx0, nx1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float pa0 = a0[j], pa1 = a1[j], pw0 = w0[j], pw1 = w1[j];
              a0[j] = c * pa0 - sn * b0[j]; nb0[j] = sn * pa0 + c * b0[j];
              a1[j] = c * pa1 - sn * b1[j]; nb1[j] = sn * pa1 + c * b1[j];
              w0[j] = c * pw0 - sn * x0[j]; nx0[j] = sn * pw0 + c * x0[j];
              w1[j] = c * pw1 - sn * x1[j]; nx1[j] = sn * pw1 + c * x1[j];
            }
            *reinterpret_cast<f32x4*>(gq) = nb0; *reinterpret_cast<f32x4*>(gq + 64) = nb1;
            *reinterpret_cast<f32x4*>(vq) = nx0; *reinterpret_cast<f32x4*>(vq + 64) = nx1;
            rotated = fmaxf(rotated, ab * ab * __builtin_amdgcn_rcpf(aa * bb));
            // |g_p|^2 after the rotation (exact arithmetic: aa - t * ab); recomputed from the
            // registers so that it does not drift over the rounds of the level
            float na = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) na += a0[j] * a0[j] + a1[j] * a1[j];
            aa = row16_sum(na);
          }
        }
        __syncthreads();
      }
      *reinterpret_cast<f32x4*>(gp) = a0; *reinterpret_cast<f32x4*>(gp + 64) = a1;
      *reinterpret_cast<f32x4*>(vp) = w0; *reinterpret_cast<f32x4*>(vp + 64) = w1;
      __syncthreads();
    }
    if (rotated > 0.f && l == 0) atomicMax(&s_rot[sweeps & 1], __float_as_int(rotated));
    __syncthreads();
    const int any = s_rot[sweeps & 1];
    __syncthreads();
    if (tid == 0) s_rot[sweeps & 1] = 0;
    if (!any || __int_as_float(any) < done_cos2) { ++sweeps_total; break; }
  }
  __syncthreads();
  return sweeps_total;
}

// ---- Q = eigenvectors of the Gram matrix of the pair: one-sided Jacobi on (P, I) in LDS (the
// solver of eigh_small_kernel), all inner sweeps in the launch.  Pairs whose largest scaled
// off-diagonal entry is already below `tol` are skipped (flag for the update kernel); the
// maximum over the sweep is what the outer iteration stops on.
__global__ __launch_bounds__(SE_T) void cj_pivot_kernel(EighBlock* blocks, const ETile* tiles,
                                                        int round, float tol, int max_inner,
                                                        float done_cos2, int sort,
                                                        int stationary, float one_below,
                                                        int store_transposed) {
  extern __shared__ __align__(16) float sem[];
  float* Gs = sem;                      // [128][132] column-major
  float* Vs = sem + SE_MAXN * SE_LD;
  float* s_red = sem + 2 * SE_MAXN * SE_LD;                // [16]
  int* s_rot = reinterpret_cast<int*>(s_red + 16);         // [2]
  float* s_dinv = s_red + 32;                              // [128]
  const ETile te = tiles[blockIdx.x];
  EighBlock* eb = &blocks[te.block];
  if (!eb->cj_active || round >= eb->nb - 1) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* P = eb->V + (int64_t)te.k * JP * JP;
  int* flags = reinterpret_cast<int*>(eb->offpart);
  if (tid < JP) s_dinv[tid] = rsqrtf(fmaxf(gload1(P + tid * JP + tid), 1e-37f));
  if (tid < 2) s_rot[tid] = 0;
  __syncthreads();
  unsigned so = 0;
  for (int e = tid; e < JP * JP; e += SE_T) {
    const int row = e >> 7, col = e & 127;
    const float p = gload1(P + (row <= col ? row * JP + col : col * JP + row));   // upper triangle
    Gs[col * SE_LD + row] = p;
    Vs[col * SE_LD + row] = row == col ? 1.f : 0.f;
    if (row != col) {
      const unsigned b = __float_as_uint(fabsf(p) * s_dinv[row] * s_dinv[col]);
      so = b > so ? b : so;
    }
  }
  so = wave_max_u32(so);
  if (lane == 0) reinterpret_cast<unsigned*>(s_red)[wave] = so;
  __syncthreads();
  unsigned som = 0;
  for (int w = 0; w < SE_T / 64; ++w) {
    const unsigned o = reinterpret_cast<unsigned*>(s_red)[w];
    som = o > som ? o : som;
  }
  const float sof = __uint_as_float(som);
  const bool rotate = sof > tol;   // NaN: no rotation, the NaN ends the iteration
  if (tid == 0) {
    atomicMax(&eb->soff_bits, som);
    flags[round * eb->npairs + te.k] = rotate ? 1 : 0;
  }
  if (!rotate) return;
  __syncthreads();
  // a pair that is already nearly orthogonal (quadratic regime) gets a single inner sweep
  if (sof < one_below) max_inner = 1;
  if (stationary) onesided_jacobi_lds_stationary(Gs, Vs, s_rot, max_inner, done_cos2);
  else onesided_jacobi_lds_t<4>(Gs, Vs, s_rot, JP, max_inner, done_cos2);
  // the approximate rcp / rsq of the rotation parameters scale a rotation by 1 + O(eps):
  // renormalise; |g_j| = |P v_j| = the eigenvalue of column j
  float* s_ev = s_dinv + JP;                              // [128]
  int* s_rank = reinterpret_cast<int*>(s_ev + JP);        // [128]
  for (int j = wave; j < JP; j += SE_T / 64) {
    const float v0 = Vs[j * SE_LD + lane], v1 = Vs[j * SE_LD + 64 + lane];
    const float g0 = Gs[j * SE_LD + lane], g1 = Gs[j * SE_LD + 64 + lane];
    const float nn = wave_sum_f32(v0 * v0 + v1 * v1);
    const float gg = wave_sum_f32(g0 * g0 + g1 * g1);
    const float inv = nn > 0.f ? 1.f / sqrtf(nn) : 0.f;
    Vs[j * SE_LD + lane] = v0 * inv; Vs[j * SE_LD + 64 + lane] = v1 * inv;
    if (lane == 0) s_ev[j] = gg;
  }
  __syncthreads();
  // Columns leave the pivot sorted by descending norm (de Rijk's ordering: without it the
  // block iteration needs 11 instead of 8 sweeps at n = 1024, and with a random order it does
  // not converge in 24: tools/proto_onesided_chol.py)
  if (tid < JP) {
    const float mine = s_ev[tid];
    int rank = 0;
    for (int i = 0; i < JP; ++i) {
      const float o = s_ev[i];
      rank += (o > mine || (o == mine && i < tid)) ? 1 : 0;
    }
    if (mine != mine) rank = tid;
    s_rank[tid] = sort ? rank : tid;
  }
  __syncthreads();
  float* Qg = eb->Q + (int64_t)te.k * JP * JP;
  if (store_transposed) {
    // Q^T[rank_j][k]: the k-contiguous right operand of cj_update_x6_kernel; rows of 512 bytes
    for (int e = tid; e < JP * JP; e += SE_T) {
      const int jc = e >> 7, k = e & 127;
      gstore1(Qg + s_rank[jc] * JP + k, Vs[jc * SE_LD + k]);
    }
    return;
  }
  for (int e = tid; e < JP * JP; e += SE_T) {
    const int k = e >> 7, jc = e & 127;          // Q[k][rank_j] = component k of eigenvector j
    gstore1(Qg + k * JP + s_rank[jc], Vs[jc * SE_LD + k]);
  }
}

// ---- [G_I G_J] <- [G_I G_J] Q on 128-row tiles (K = 128 gathered from the two block columns)
template <int UBK>
__global__ __launch_bounds__(256, 2) void cj_update_kernel_t(EighBlock* blocks, const ETile* tiles,
                                                             int ntiles, int round) {
  __shared__ __align__(16) float smem[SmemCfg<UBK>::total(KC, MC)];
  const ETile te = tiles[xcd_remap(blockIdx.x, ntiles)];
  EighBlock* eb = &blocks[te.block];
  if (!eb->cj_active || round >= eb->nb - 1) return;
  if (reinterpret_cast<const int*>(eb->offpart)[round * eb->npairs + te.k] == 0) return;
  int I, J;
  rr_pair(eb->nb, round, te.k, I, J);
  const int ld = eb->npad;
  const float* Qg = eb->Q + (int64_t)te.k * JP * JP;
  float* X = eb->X;
  f32x16 acc[2][2];
  zero_acc(acc);
  for (int seg = 0; seg < 2; ++seg) {
    const int cb = (seg == 0 ? I : J) * JB;
    Operand a{X + cb, ld, te.t * TILE, ld, JB, true};         // (m,k) = X[rt+m][cb+k]
    Operand b{Qg + seg * JB * JP, JP, 0, JP, JB, true};       // (j,k) = Q[seg*64+k][j]
    gemm_tile_accum<KC, MC, UBK, false>(a, b, JB, smem, acc);
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = te.t * TILE + acc_row(wm, i, r, lane);
        const int c = acc_col(wn, j, lane);
        const int col = c < JB ? I * JB + c : J * JB + (c - JB);
        gstore1(X + (int64_t)row * ld + col, acc[i][j][r]);
      }
}

// ---- the same update on the bf16 MFMA: both operands split three ways (gemm_bf16x.hip.h), six
// partial products per product, float32 accumulation: ~2^-22 relative per product -- the size of
// the float32 chain's own rounding at K = 128 -- at 6 / 16 of the float32 MFMA time, which turns
// the update from an MFMA-bound into an HBM-bound kernel (it reads and writes G once per round)
// and leaves the shared VALU / MFMA pipe to the pivot kernel of the other stream group.  The
// rotation Q is orthogonal to float32 accuracy whatever the arithmetic of its application, so
// G G^T = D holds to the product's rounding as before; the convergence test stays on the float32
// Gram kernel.  Q arrives transposed (cj_pivot_kernel store_transposed): Q^T[c][k].
__global__ __launch_bounds__(256, 2) void cj_update_x6_kernel(EighBlock* blocks, const ETile* tiles,
                                                              int ntiles, int round) {
  extern __shared__ __align__(16) float smem_x6[];   // 6 planes of 128 x 40 bf16 = 61,440 bytes
  const ETile te = tiles[xcd_remap(blockIdx.x, ntiles)];
  EighBlock* eb = &blocks[te.block];
  if (!eb->cj_active || round >= eb->nb - 1) return;
  if (reinterpret_cast<const int*>(eb->offpart)[round * eb->npairs + te.k] == 0) return;
  int I, J;
  rr_pair(eb->nb, round, te.k, I, J);
  const int ld = eb->npad;
  const float* Qt = eb->Q + (int64_t)te.k * JP * JP;
  float* X = eb->X;
  f32x16 acc[2][2];
  // (m, k) = X[rt + m][cb + k], (c, k) = Q^T[c][seg * 64 + k]
  gemm_tile_bf16x_sym<6, false>(X + I * JB, ld, te.t * TILE, Qt, JP, 0, JB, smem_x6, acc);
  gemm_tile_bf16x_sym<6, true>(X + J * JB, ld, te.t * TILE, Qt + JB, JP, 0, JB, smem_x6, acc);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = te.t * TILE + acc_row(wm, i, r, lane);
        const int c = acc_col(wn, j, lane);
        const int col = c < JB ? I * JB + c : J * JB + (c - JB);
        gstore1(X + (int64_t)row * ld + col, acc[i][j][r]);
      }
}

// ---- per-sweep control: blocks whose largest scaled Gram entry of the sweep is under `tol`
// are done.  mode 0 = after the Cholesky factorisation (failed blocks go to the fallback).
__global__ __launch_bounds__(256) void cj_control_kernel(EighBlock* blocks, int nblocks, int mode,
                                                         float tol, int gen, EStatus* status,
                                                         int group) {
  __shared__ int s_act, s_fail;
  __shared__ float s_max;
  if (threadIdx.x == 0) { s_act = 0; s_fail = 0; s_max = 0.f; }
  __syncthreads();
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    EighBlock* eb = &blocks[b];
    if (!eb->cj || (group >= 0 && eb->cj_group != group)) continue;
    if (mode == 0) {
      if (eb->chol_fail) { eb->cj = 0; eb->cj_active = 0; eb->active = 1; atomicAdd(&s_fail, 1); }
    } else if (eb->cj_active) {
      const float so = __uint_as_float(eb->soff_bits);
      eb->soff_bits = 0;
      eb->off_rel = so;
      eb->sweeps += 1;
      if (!(so > tol)) eb->cj_active = 0;
    }
    if (eb->cj_active) {
      atomicAdd(&s_act, 1);
      atomicMax(reinterpret_cast<int*>(&s_max), __float_as_int(fmaxf(eb->off_rel, 0.f)));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && status) {
    status->active = s_act;
    status->max_off = s_max;
    status->pad_ = s_fail;
    __threadfence_system();
    status->gen = gen;
  }
}

// ---- eigenvalues |g_j|^2 (float64 accumulation) of a chunk of 128 columns
__global__ __launch_bounds__(256) void cj_norms_kernel(EighBlock* blocks, const ETile* tiles) {
  __shared__ double part[2][TILE];
  const ETile te = tiles[blockIdx.x];    // sq tile list: only te.k == 0 tiles work, te.t = chunk
  EighBlock* eb = &blocks[te.block];
  if (!eb->cj || te.k != 0) return;
  const int ld = eb->npad, tid = threadIdx.x;
  const float* G = eb->X;
  const int c = te.t * TILE + (tid & 127), half = tid >> 7;
  double s = 0.0;
  for (int r = half; r < ld; r += 2) {
    const double g = (double)gload1(G + (int64_t)r * ld + c);
    s = fma(g, g, s);
  }
  part[half][tid & 127] = s;
  __syncthreads();
  if (tid < TILE) eb->evals[te.t * TILE + tid] = (float)(part[0][tid] + part[1][tid]);
}

// ---- Rayleigh quotients e_j = u_j^T (D u_j) from the float64-accumulated product D U kept as a
// float32 hi/lo pair in (X, W) by eigh_reproject_f64_kernel<0>: diag(A) = e (float64 sum).
__global__ __launch_bounds__(256) void cj_rayleigh_from_dv_kernel(EighBlock* blocks,
                                                                  const ETile* tiles) {
  __shared__ double part[2][TILE];
  const ETile te = tiles[blockIdx.x];    // sq tile list: only te.k == 0 tiles work, te.t = chunk
  EighBlock* eb = &blocks[te.block];
  if (!eb->cj || te.k != 0) return;
  const int ld = eb->npad, n = eb->n, tid = threadIdx.x;
  const int c = te.t * TILE + (tid & 127), half = tid >> 7;
  double s = 0.0;
  for (int r = half; r < n; r += 2) {
    const int64_t o = (int64_t)r * ld + c;
    s = fma((double)gload1(eb->V + o), (double)gload1(eb->X + o) + (double)gload1(eb->W + o), s);
  }
  part[half][tid & 127] = s;
  __syncthreads();
  if (tid < TILE && c < n) eb->A[(int64_t)c * ld + c] = (float)(part[0][tid] + part[1][tid]);
}

// ---- V = G diag(1 / |g_j|), diag(A) = |g_j|^2
__global__ __launch_bounds__(256) void cj_finalize_kernel(EighBlock* blocks, const ETile* tiles) {
  __shared__ float inv[TILE];
  const ETile te = tiles[blockIdx.x];
  EighBlock* eb = &blocks[te.block];
  if (!eb->cj) return;
  const int ld = eb->npad, n = eb->n, tid = threadIdx.x;
  if (tid < TILE) {
    const int c = te.t * TILE + tid;
    const float e = eb->evals[c];
    inv[tid] = (c < n && e > 0.f) ? (float)(1.0 / sqrt((double)e)) : (e == e ? 0.f : e);
  }
  __syncthreads();
  const float* G = eb->X;
  for (int e = tid; e < TILE * TILE; e += 256) {
    const int r = e >> 7, c = e & 127;
    const int row = te.k * TILE + r, col = te.t * TILE + c;
    const int64_t o = (int64_t)row * ld + col;
    float v = gload1(G + o) * inv[c];
    if (row >= n || col >= n) v = row == col ? 1.f : 0.f;
    eb->V[o] = v;
    eb->A[o] = row == col ? (col < n ? eb->evals[col] : 0.f) : 0.f;
  }
}
