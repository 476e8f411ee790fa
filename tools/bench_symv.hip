// Dev (round 5): what limits the symmetric mat-vec of the tridiagonal reduction (td_symv_kernel)?
// Standalone micro-benchmark: 64 matrices of 2048^2 floats, one step over the upper block triangle.
//   v0  the kernel's shape: one 128 x 128 tile per workgroup, row sums by shuffles, column sums via LDS
//   v1  the same tile loads, no arithmetic beyond one sum (what this access pattern can stream)
//   v2  128 x 128 tiles, two per workgroup (the second tile's loads in flight during the first's sums)
//   v3  row strips: a workgroup owns 32 rows x the whole width right of the diagonal (2 KB+ runs)
// build: hipcc -O3 --offload-arch=gfx950 tools/bench_symv.hip -o tools/bench_symv
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define G1 __attribute__((address_space(1)))
__device__ inline f32x4 gload4(const float* p) { return *(const f32x4 G1*)(p); }

constexpr int N = 2048, NT = 16, TILE = 128, NB = 64;

__device__ inline void decode(int q, int T, int& I, int& J) {
  int Ip = 0;
  while (q >= T - Ip) { q -= T - Ip; ++Ip; }
  I = Ip; J = Ip + q;
}

__global__ __launch_bounds__(256) void v0(const float* A, const float* v, float* slab) {
  __shared__ float svI[TILE], svJ[TILE], srow[TILE], scol[8][TILE];
  const float* a_ = A + (size_t)blockIdx.y * N * N;
  int I, J; decode(blockIdx.x, NT, I, J);
  const int tid = threadIdx.x, r0 = tid >> 5, c4 = (tid & 31) * 4;
  const float* src = a_ + (size_t)(I * TILE + r0) * N + J * TILE + c4;
  f32x4 a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = gload4(src + (size_t)(8 * k) * N);
  if (tid < TILE) { svI[tid] = v[blockIdx.y * N + I * TILE + tid]; svJ[tid] = v[blockIdx.y * N + J * TILE + tid]; }
  __syncthreads();
  const f32x4 vj = *reinterpret_cast<const f32x4*>(&svJ[c4]);
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float rs = a[k][0] * vj[0] + a[k][1] * vj[1] + a[k][2] * vj[2] + a[k][3] * vj[3];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
    if ((tid & 31) == 0) srow[r0 + 8 * k] = rs;
    const float vi = svI[r0 + 8 * k];
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[e] += a[k][e] * vi;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) scol[r0][c4 + e] = cs[e];
  __syncthreads();
  float* sl = slab + (size_t)blockIdx.y * NT * NT * TILE;
  if (tid < TILE) {
    sl[(I * NT + J) * TILE + tid] = srow[tid];
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += scol[g][tid];
    if (I != J) sl[(J * NT + I) * TILE + tid] = s;
  }
}

__global__ __launch_bounds__(256, 6) void v0o6(const float* A, const float* v, float* slab) {
  __shared__ float svI[TILE], svJ[TILE], srow[TILE], scol[8][TILE];
  const float* a_ = A + (size_t)blockIdx.y * N * N;
  int I, J; decode(blockIdx.x, NT, I, J);
  const int tid = threadIdx.x, r0 = tid >> 5, c4 = (tid & 31) * 4;
  const float* src = a_ + (size_t)(I * TILE + r0) * N + J * TILE + c4;
  f32x4 a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = gload4(src + (size_t)(8 * k) * N);
  if (tid < TILE) { svI[tid] = v[blockIdx.y * N + I * TILE + tid]; svJ[tid] = v[blockIdx.y * N + J * TILE + tid]; }
  __syncthreads();
  const f32x4 vj = *reinterpret_cast<const f32x4*>(&svJ[c4]);
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float rs = a[k][0] * vj[0] + a[k][1] * vj[1] + a[k][2] * vj[2] + a[k][3] * vj[3];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
    if ((tid & 31) == 0) srow[r0 + 8 * k] = rs;
    const float vi = svI[r0 + 8 * k];
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[e] += a[k][e] * vi;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) scol[r0][c4 + e] = cs[e];
  __syncthreads();
  float* sl = slab + (size_t)blockIdx.y * NT * NT * TILE;
  if (tid < TILE) {
    sl[(I * NT + J) * TILE + tid] = srow[tid];
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += scol[g][tid];
    if (I != J) sl[(J * NT + I) * TILE + tid] = s;
  }
}


__global__ __launch_bounds__(256, 8) void v0o8(const float* A, const float* v, float* slab) {
  __shared__ float svI[TILE], svJ[TILE], srow[TILE], scol[8][TILE];
  const float* a_ = A + (size_t)blockIdx.y * N * N;
  int I, J; decode(blockIdx.x, NT, I, J);
  const int tid = threadIdx.x, r0 = tid >> 5, c4 = (tid & 31) * 4;
  const float* src = a_ + (size_t)(I * TILE + r0) * N + J * TILE + c4;
  f32x4 a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = gload4(src + (size_t)(8 * k) * N);
  if (tid < TILE) { svI[tid] = v[blockIdx.y * N + I * TILE + tid]; svJ[tid] = v[blockIdx.y * N + J * TILE + tid]; }
  __syncthreads();
  const f32x4 vj = *reinterpret_cast<const f32x4*>(&svJ[c4]);
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float rs = a[k][0] * vj[0] + a[k][1] * vj[1] + a[k][2] * vj[2] + a[k][3] * vj[3];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
    if ((tid & 31) == 0) srow[r0 + 8 * k] = rs;
    const float vi = svI[r0 + 8 * k];
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[e] += a[k][e] * vi;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) scol[r0][c4 + e] = cs[e];
  __syncthreads();
  float* sl = slab + (size_t)blockIdx.y * NT * NT * TILE;
  if (tid < TILE) {
    sl[(I * NT + J) * TILE + tid] = srow[tid];
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += scol[g][tid];
    if (I != J) sl[(J * NT + I) * TILE + tid] = s;
  }
}


__global__ __launch_bounds__(256) void v0a(const float* A, const float* v, float* slab) {
  __shared__ float svI[TILE], svJ[TILE], srow[TILE], scol[8][TILE];
  const float* a_ = A + (size_t)blockIdx.y * N * N;
  int I, J; decode(blockIdx.x, NT, I, J);
  const int tid = threadIdx.x, r0 = tid >> 5, c4 = (tid & 31) * 4;
  const float* src = a_ + (size_t)(I * TILE + r0) * N + J * TILE + c4;
  f32x4 a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = gload4(src + (size_t)(8 * k) * N);
  if (tid < TILE) { svI[tid] = v[blockIdx.y * N + I * TILE + tid]; svJ[tid] = v[blockIdx.y * N + J * TILE + tid]; }
  __syncthreads();
  const f32x4 vj = *reinterpret_cast<const f32x4*>(&svJ[c4]);
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float rs = a[k][0] * vj[0] + a[k][1] * vj[1] + a[k][2] * vj[2] + a[k][3] * vj[3];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
    if ((tid & 31) == 0) srow[r0 + 8 * k] = rs;
    const float vi = svI[r0 + 8 * k];
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[e] += a[k][e] * vi;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) scol[r0][c4 + e] = cs[e];
  __syncthreads();
  float* sl = slab + (size_t)blockIdx.y * NT * NT * TILE;
  if (tid < TILE) {
    float s = srow[tid];
#pragma unroll
    for (int g = 0; g < 8; ++g) s += scol[g][tid];
    if (s == 12345.f) sl[(J * NT + I) * TILE + tid] = s;   // no stores
  }
}


__global__ __launch_bounds__(256) void v0c(const float* A, const float* v, float* slab) {
  __shared__ float svI[TILE], svJ[TILE], srow[TILE], scol[8][TILE];
  const float* a_ = A + (size_t)blockIdx.y * N * N;
  for (int rep = 0; rep < 2; ++rep) {
  int I, J; decode(blockIdx.x * 2 + rep, NT, I, J);
  __syncthreads();
  const int tid = threadIdx.x, r0 = tid >> 5, c4 = (tid & 31) * 4;
  const float* src = a_ + (size_t)(I * TILE + r0) * N + J * TILE + c4;
  f32x4 a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = gload4(src + (size_t)(8 * k) * N);
  if (tid < TILE) { svI[tid] = v[blockIdx.y * N + I * TILE + tid]; svJ[tid] = v[blockIdx.y * N + J * TILE + tid]; }
  __syncthreads();
  const f32x4 vj = *reinterpret_cast<const f32x4*>(&svJ[c4]);
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float rs = a[k][0] * vj[0] + a[k][1] * vj[1] + a[k][2] * vj[2] + a[k][3] * vj[3];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
    if ((tid & 31) == 0) srow[r0 + 8 * k] = rs;
    const float vi = svI[r0 + 8 * k];
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[e] += a[k][e] * vi;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) scol[r0][c4 + e] = cs[e];
  __syncthreads();
  float* sl = slab + (size_t)blockIdx.y * NT * NT * TILE;
  if (tid < TILE) {
    sl[(I * NT + J) * TILE + tid] = srow[tid];
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += scol[g][tid];
    if (I != J) sl[(J * NT + I) * TILE + tid] = s;
  }
  }
}

// v5: 16 lanes per row piece (64 columns), row sums by four DPP rotate-adds inside a row of 16 lanes
template <int N>
__device__ inline float row16_add_ror(float x) {
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + N, 0xf, 0xf, false);
  return x + __int_as_float(r);
}
__device__ inline float row16_sum(float x) {
  x = row16_add_ror<8>(x); x = row16_add_ror<4>(x); x = row16_add_ror<2>(x);
  return row16_add_ror<1>(x);
}
__global__ __launch_bounds__(256) void v5(const float* A, const float* v, float* slab) {
  __shared__ float svI[TILE], svJ[TILE], srow[TILE], scol[16][TILE];
  const float* a_ = A + (size_t)blockIdx.y * N * N;
  int I, J; decode(blockIdx.x, NT, I, J);
  const int tid = threadIdx.x, rg = tid >> 4, l16 = tid & 15;
  const float* src = a_ + (size_t)(I * TILE + rg) * N + J * TILE + l16 * 4;
  f32x4 a[8][2];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k][0] = gload4(src + (size_t)(16 * k) * N);
    a[k][1] = gload4(src + (size_t)(16 * k) * N + 64);
  }
  if (tid < TILE) { svI[tid] = v[blockIdx.y * N + I * TILE + tid]; svJ[tid] = v[blockIdx.y * N + J * TILE + tid]; }
  __syncthreads();
  const f32x4 vj0 = *reinterpret_cast<const f32x4*>(&svJ[l16 * 4]);
  const f32x4 vj1 = *reinterpret_cast<const f32x4*>(&svJ[64 + l16 * 4]);
  float cs[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float rs = a[k][0][0] * vj0[0] + a[k][0][1] * vj0[1] + a[k][0][2] * vj0[2] + a[k][0][3] * vj0[3] +
               a[k][1][0] * vj1[0] + a[k][1][1] * vj1[1] + a[k][1][2] * vj1[2] + a[k][1][3] * vj1[3];
    rs = row16_sum(rs);
    if (l16 == 0) srow[rg + 16 * k] = rs;
    const float vi = svI[rg + 16 * k];
#pragma unroll
    for (int e = 0; e < 4; ++e) { cs[0][e] += a[k][0][e] * vi; cs[1][e] += a[k][1][e] * vi; }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { scol[rg][l16 * 4 + e] = cs[0][e]; scol[rg][64 + l16 * 4 + e] = cs[1][e]; }
  __syncthreads();
  float* sl = slab + (size_t)blockIdx.y * NT * NT * TILE;
  if (tid < TILE) {
    sl[(I * NT + J) * TILE + tid] = srow[tid];
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += scol[g][tid];
    if (I != J) sl[(J * NT + I) * TILE + tid] = s;
  }
}

__global__ __launch_bounds__(256) void v1(const float* A, const float* v, float* slab) {
  const float* a_ = A + (size_t)blockIdx.y * N * N;
  int I, J; decode(blockIdx.x, NT, I, J);
  const int tid = threadIdx.x, r0 = tid >> 5, c4 = (tid & 31) * 4;
  const float* src = a_ + (size_t)(I * TILE + r0) * N + J * TILE + c4;
  f32x4 a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = gload4(src + (size_t)(8 * k) * N);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += a[k][0] + a[k][1] + a[k][2] + a[k][3];
  if (s == 12345.f) slab[blockIdx.x] = s;
}

// rows-contiguous variant of v1: thread -> 16 consecutive float4 of ONE row (a wave reads 64 rows x 16 B
// per instruction: the opposite extreme, bad coalescing per instruction but 256 B per lane in a row)
__global__ __launch_bounds__(256) void v1b(const float* A, const float* v, float* slab) {
  const float* a_ = A + (size_t)blockIdx.y * N * N;
  int I, J; decode(blockIdx.x, NT, I, J);
  const int tid = threadIdx.x, r = tid >> 1, half = tid & 1;
  const float* src = a_ + (size_t)(I * TILE + r) * N + J * TILE + half * 64;
  f32x4 a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = gload4(src + 4 * k);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += a[k][0] + a[k][1] + a[k][2] + a[k][3];
  if (s == 12345.f) slab[blockIdx.x] = s;
}

// v3: strips of 32 rows: workgroup (strip s) reads rows [32 s, 32 s + 32) x columns [128 * (s / 4), N):
// a wave reads 1 row x 1 KB per instruction, consecutive instructions walk along the row
__global__ __launch_bounds__(256) void v3(const float* A, const float* v, float* slab) {
  const float* a_ = A + (size_t)blockIdx.y * N * N;
  // pair strips s and 63 - s for balance: blockIdx.x in [0, 32)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float s = 0.f;
  for (int half = 0; half < 2; ++half) {
    const int strip = half == 0 ? blockIdx.x : 63 - blockIdx.x;
    const int c0 = (strip / 4) * TILE;
    for (int rr = wave; rr < 32; rr += 4) {
      const float* row = a_ + (size_t)(strip * 32 + rr) * N;
      for (int c = c0 + lane * 4; c < N; c += 1024) {
        f32x4 x0 = gload4(row + c);
        f32x4 x1 = c + 256 < N ? gload4(row + c + 256) : f32x4{0, 0, 0, 0};
        f32x4 x2 = c + 512 < N ? gload4(row + c + 512) : f32x4{0, 0, 0, 0};
        f32x4 x3 = c + 768 < N ? gload4(row + c + 768) : f32x4{0, 0, 0, 0};
        s += x0[0] + x1[1] + x2[2] + x3[3];
      }
    }
  }
  if (s == 12345.f) slab[blockIdx.x] = s;
}

// v4: strips of 8 rows per wave, 4 rows in flight per lane group: each workgroup takes a 32-row strip
// but every thread issues 16 loads up front (4 rows x 4 segments of 1 KB)
__global__ __launch_bounds__(256) void v4(const float* A, const float* v, float* slab) {
  const float* a_ = A + (size_t)blockIdx.y * N * N;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float s = 0.f;
  for (int half = 0; half < 2; ++half) {
    const int strip = half == 0 ? blockIdx.x : 63 - blockIdx.x;
    const int c0 = (strip / 4) * TILE;
    for (int cb = c0; cb < N; cb += 1024) {
      f32x4 x[8][4];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float* row = a_ + (size_t)(strip * 32 + wave * 8 + r) * N + cb + lane * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) x[r][q] = cb + q * 256 + lane * 4 < N ? gload4(row + q * 256) : f32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) s += x[r][q][0] + x[r][q][3];
    }
  }
  if (s == 12345.f) slab[blockIdx.x] = s;
}


// v0d: v0 behind the dependent chain of the real kernel: block descriptor -> tile loads, then the Householder
// scalars (16 float64 partial sums + alpha) -> the vector scaled by them.  v0e: descriptor only.
struct Desc { int n, ld, nt, pad; const float* A; const float* u; const double* pss; float* slab; };
template <int HOUSE>
__global__ __launch_bounds__(256) void v0d(const Desc* descs, int j) {
  __shared__ float svI[TILE], svJ[TILE], srow[TILE], scol[8][TILE];
  const Desc* d = &descs[blockIdx.y];
  const int ld = d->ld, nt = d->nt;
  const float* a_ = d->A;
  int I, J; decode(blockIdx.x, nt, I, J);
  const int tid = threadIdx.x, r0 = tid >> 5, c4 = (tid & 31) * 4;
  const float* src = a_ + (size_t)(I * TILE + r0) * ld + J * TILE + c4;
  f32x4 a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = gload4(src + (size_t)(8 * k) * ld);
  float scale = 1.f;
  float uI = 0.f, uJ = 0.f;
  if (HOUSE >= 3 && tid < TILE) { uI = d->u[I * TILE + tid]; uJ = d->u[J * TILE + tid]; }   // before the scalars: one round trip
  if (HOUSE) {
    double sigma = 0.0;
    if (HOUSE == 1) {
      for (int x = j / TILE; x < nt; ++x) sigma += d->pss[x];
    } else if (HOUSE < 4) {   // all partial sums requested at once (same summation order: masked entries add 0.0)
      double ps[32];
      const int x0 = j / TILE;
#pragma unroll
      for (int x = 0; x < 32; ++x) ps[x] = (x >= x0 && x < nt) ? d->pss[x] : 0.0;
#pragma unroll
      for (int x = 0; x < 32; ++x) sigma += ps[x];
    } else {   // the pointer read once, 32 unconditional loads (scalar: uniform addresses), masked in the sum
      const double* __restrict__ pss = d->pss;
      double ps[32];
      const int x0 = j / TILE;
#pragma unroll
      for (int x = 0; x < 32; ++x) ps[x] = pss[x];
#pragma unroll
      for (int x = 0; x < 32; ++x) sigma += (x >= x0 && x < nt) ? ps[x] : 0.0;
    }
    const float alpha = d->u[j + 1];
    const float sf = (float)sigma, a2 = alpha * alpha;
    const float beta = -copysignf(sqrtf(a2 + sf), alpha);
    scale = 1.f / (alpha - beta);
  }
  if (tid < TILE) {
    if (HOUSE >= 3) { svI[tid] = uI * scale; svJ[tid] = uJ * scale; }
    else { svI[tid] = d->u[I * TILE + tid] * scale; svJ[tid] = d->u[J * TILE + tid] * scale; }
  }
  __syncthreads();
  const f32x4 vj = *reinterpret_cast<const f32x4*>(&svJ[c4]);
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float rs = a[k][0] * vj[0] + a[k][1] * vj[1] + a[k][2] * vj[2] + a[k][3] * vj[3];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
    if ((tid & 31) == 0) srow[r0 + 8 * k] = rs;
    const float vi = svI[r0 + 8 * k];
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[e] += a[k][e] * vi;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) scol[r0][c4 + e] = cs[e];
  __syncthreads();
  float* sl = d->slab;
  if (tid < TILE) {
    sl[(I * nt + J) * TILE + tid] = srow[tid];
    float s2 = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s2 += scol[g][tid];
    if (I != J) sl[(J * nt + I) * TILE + tid] = s2;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  float *A, *v, *slab;
  const size_t bytes = (size_t)NB * N * N * sizeof(float);
  CK(hipMalloc(&A, bytes)); CK(hipMalloc(&v, NB * N * sizeof(float))); CK(hipMalloc(&slab, (size_t)NB * NT * NT * TILE * 4));
  {  // random data: an all-zero matrix draws less power and flatters the clocks
    std::vector<float> h((size_t)N * N);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    for (int b = 0; b < NB; ++b) CK(hipMemcpy(A + (size_t)b * N * N, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(v, h.data(), NB * N * 4, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int ntri = NT * (NT + 1) / 2;
  const double tri_bytes = (double)NB * ntri * TILE * TILE * 4;
  auto time = [&](const char* name, auto launch, double by) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-6s %8.1f us per launch  %6.2f TB/s\n", name, ms * 1e3 / reps, by / (ms * 1e-3 / reps) / 1e12);
  };
  time("v0", [&] { hipLaunchKernelGGL(v0, dim3(ntri, NB), dim3(256), 0, 0, A, v, slab); }, tri_bytes);
  for (int T : {12, 8, 4, 2, 1}) {   // later steps of the reduction: fewer tiles per launch (the tiles read are the first T rows' worth: same shape)
    char nm[16]; snprintf(nm, sizeof nm, "v0T%d", T);
    const int nt_ = T * (T + 1) / 2;
    time(nm, [&] { hipLaunchKernelGGL(v0, dim3(nt_, NB), dim3(256), 0, 0, A, v, slab); }, (double)NB * nt_ * TILE * TILE * 4);
  }
  {
    Desc hd[NB]; Desc* dd; double* pss; CK(hipMalloc(&dd, sizeof(hd))); CK(hipMalloc(&pss, NB * 32 * 8));
    { double hp[NB * 32]; for (int i = 0; i < NB * 32; ++i) hp[i] = 1.0 + i; CK(hipMemcpy(pss, hp, sizeof(hp), hipMemcpyHostToDevice)); }
    for (int b = 0; b < NB; ++b)
      hd[b] = Desc{N, N, NT, 0, A + (size_t)b * N * N, v + (size_t)b * N, pss + b * 32, slab + (size_t)b * NT * NT * TILE};
    CK(hipMemcpy(dd, hd, sizeof(hd), hipMemcpyHostToDevice));
    time("v0e", [&] { hipLaunchKernelGGL(v0d<0>, dim3(ntri, NB), dim3(256), 0, 0, dd, 5); }, tri_bytes);
    time("v0d", [&] { hipLaunchKernelGGL(v0d<1>, dim3(ntri, NB), dim3(256), 0, 0, dd, 5); }, tri_bytes);
    time("v0f", [&] { hipLaunchKernelGGL(v0d<2>, dim3(ntri, NB), dim3(256), 0, 0, dd, 5); }, tri_bytes);
    time("v0", [&] { hipLaunchKernelGGL(v0, dim3(ntri, NB), dim3(256), 0, 0, A, v, slab); }, tri_bytes);
    time("v0e", [&] { hipLaunchKernelGGL(v0d<0>, dim3(ntri, NB), dim3(256), 0, 0, dd, 5); }, tri_bytes);
    time("v0d", [&] { hipLaunchKernelGGL(v0d<1>, dim3(ntri, NB), dim3(256), 0, 0, dd, 5); }, tri_bytes);
    time("v0f", [&] { hipLaunchKernelGGL(v0d<2>, dim3(ntri, NB), dim3(256), 0, 0, dd, 5); }, tri_bytes);
    time("v0g", [&] { hipLaunchKernelGGL(v0d<3>, dim3(ntri, NB), dim3(256), 0, 0, dd, 5); }, tri_bytes);
    time("v0h", [&] { hipLaunchKernelGGL(v0d<4>, dim3(ntri, NB), dim3(256), 0, 0, dd, 5); }, tri_bytes);
    time("v0h", [&] { hipLaunchKernelGGL(v0d<4>, dim3(ntri, NB), dim3(256), 0, 0, dd, 5); }, tri_bytes);
  }
  time("v0o6", [&] { hipLaunchKernelGGL(v0o6, dim3(ntri, NB), dim3(256), 0, 0, A, v, slab); }, tri_bytes);
  time("v0o8", [&] { hipLaunchKernelGGL(v0o8, dim3(ntri, NB), dim3(256), 0, 0, A, v, slab); }, tri_bytes);
  time("v0a", [&] { hipLaunchKernelGGL(v0a, dim3(ntri, NB), dim3(256), 0, 0, A, v, slab); }, tri_bytes);
  time("v0c", [&] { hipLaunchKernelGGL(v0c, dim3(ntri / 2, NB), dim3(256), 0, 0, A, v, slab); }, tri_bytes);
  time("v5", [&] { hipLaunchKernelGGL(v5, dim3(ntri, NB), dim3(256), 0, 0, A, v, slab); }, tri_bytes);
  time("v0", [&] { hipLaunchKernelGGL(v0, dim3(ntri, NB), dim3(256), 0, 0, A, v, slab); }, tri_bytes);
  time("v5", [&] { hipLaunchKernelGGL(v5, dim3(ntri, NB), dim3(256), 0, 0, A, v, slab); }, tri_bytes);
  time("v1", [&] { hipLaunchKernelGGL(v1, dim3(ntri, NB), dim3(256), 0, 0, A, v, slab); }, tri_bytes);
  time("v1b", [&] { hipLaunchKernelGGL(v1b, dim3(ntri, NB), dim3(256), 0, 0, A, v, slab); }, tri_bytes);
  // strips read the block triangle rounded to 128 columns as well
  double strip_bytes = 0;
  for (int s = 0; s < 64; ++s) strip_bytes += 32.0 * (N - (s / 4) * TILE) * 4;
  strip_bytes *= NB;
  time("v3", [&] { hipLaunchKernelGGL(v3, dim3(32, NB), dim3(256), 0, 0, A, v, slab); }, strip_bytes);
  time("v4", [&] { hipLaunchKernelGGL(v4, dim3(32, NB), dim3(256), 0, 0, A, v, slab); }, strip_bytes);
  printf("triangle bytes %.1f MB, strip bytes %.1f MB\n", tri_bytes / 1e6, strip_bytes / 1e6);
  return 0;
}
