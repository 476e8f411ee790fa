// Dev (round 5): streaming ceilings for the dequantize shape (2-byte codes in, 4-byte floats out) on
// 282.8 M elements: which access widths reach what, and what a write-heavy stream can do at all.
// build: hipcc -O3 --offload-arch=gfx950 tools/bench_stream.hip -o tools/bin/bench_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define G1 __attribute__((address_space(1)))

constexpr long long NEL = 282800128ll;   // multiple of 16384
constexpr int CH = 16384;

// s1: one 8-byte load -> one 16-byte store per lane and step (the dequantize kernel's widths), 16 steps
__global__ __launch_bounds__(256) void s1(const short* in, float* out) {
  const long long base = (long long)blockIdx.x * CH + threadIdx.x * 4;
  u32x2 v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = *(const u32x2 G1*)(in + base + 1024 * k);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    f32x4 o = {(float)(short)(v[k][0] & 0xffff), (float)(short)(v[k][0] >> 16), (float)(short)(v[k][1] & 0xffff),
               (float)(short)(v[k][1] >> 16)};
    *(f32x4 G1*)(out + base + 1024 * k) = o;
  }
}
// s2: one 16-byte load -> two adjacent 16-byte stores per lane and step, 8 steps
__global__ __launch_bounds__(256) void s2(const short* in, float* out) {
  const long long base = (long long)blockIdx.x * CH + threadIdx.x * 8;
  u32x4 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = *(const u32x4 G1*)(in + base + 2048 * k);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    f32x4 o0 = {(float)(short)(v[k][0] & 0xffff), (float)(short)(v[k][0] >> 16), (float)(short)(v[k][1] & 0xffff),
                (float)(short)(v[k][1] >> 16)};
    f32x4 o1 = {(float)(short)(v[k][2] & 0xffff), (float)(short)(v[k][2] >> 16), (float)(short)(v[k][3] & 0xffff),
                (float)(short)(v[k][3] >> 16)};
    *(f32x4 G1*)(out + base + 2048 * k) = o0;
    *(f32x4 G1*)(out + base + 2048 * k + 4) = o1;
  }
}
// s3: float4 copy (4 bytes in, 4 bytes out per element)
__global__ __launch_bounds__(256) void s3(const float* in, float* out) {
  const long long base = (long long)blockIdx.x * CH + threadIdx.x * 4;
  f32x4 v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = *(const f32x4 G1*)(in + base + 1024 * k);
#pragma unroll
  for (int k = 0; k < 16; ++k) *(f32x4 G1*)(out + base + 1024 * k) = v[k];
}
// s4: write only
__global__ __launch_bounds__(256) void s4(const float* in, float* out) {
  const long long base = (long long)blockIdx.x * CH + threadIdx.x * 4;
  const f32x4 o = {1.f, 2.f, 3.f, (float)threadIdx.x};
#pragma unroll
  for (int k = 0; k < 16; ++k) *(f32x4 G1*)(out + base + 1024 * k) = o;
}
// s5: read only (float4)
__global__ __launch_bounds__(256) void s5(const float* in, float* out) {
  const long long base = (long long)blockIdx.x * CH + threadIdx.x * 4;
  f32x4 v[16];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = *(const f32x4 G1*)(in + base + 1024 * k);
#pragma unroll
  for (int k = 0; k < 16; ++k) s += v[k][0] + v[k][3];
  if (s == 12345.f) out[blockIdx.x] = s;
}
// s6: s1 with nontemporal stores
__global__ __launch_bounds__(256) void s6(const short* in, float* out) {
  const long long base = (long long)blockIdx.x * CH + threadIdx.x * 4;
  u32x2 v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = *(const u32x2 G1*)(in + base + 1024 * k);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    f32x4 o = {(float)(short)(v[k][0] & 0xffff), (float)(short)(v[k][0] >> 16), (float)(short)(v[k][1] & 0xffff),
               (float)(short)(v[k][1] >> 16)};
    __builtin_nontemporal_store(o, (f32x4*)(out + base + 1024 * k));
  }
}

// s7: s1 + a float4 of per-column scales per step from a 4 KB table (L2-resident), multiplied in
__global__ __launch_bounds__(256) void s7(const short* in, float* out, const float* bucket) {
  const long long base = (long long)blockIdx.x * CH + threadIdx.x * 4;
  u32x2 v[16];
  f32x4 b[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    v[k] = *(const u32x2 G1*)(in + base + 1024 * k);
    b[k] = *(const f32x4 G1*)(bucket + ((threadIdx.x * 4) & 1023));
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    f32x4 o = {(float)(short)(v[k][0] & 0xffff) * b[k][0], (float)(short)(v[k][0] >> 16) * b[k][1],
               (float)(short)(v[k][1] & 0xffff) * b[k][2], (float)(short)(v[k][1] >> 16) * b[k][3]};
    *(f32x4 G1*)(out + base + 1024 * k) = o;
  }
}
// s8: s7 with the scales loaded INSIDE the conversion loop (a dependent L2 round trip per step)
__global__ __launch_bounds__(256) void s8(const short* in, float* out, const float* bucket) {
  const long long base = (long long)blockIdx.x * CH + threadIdx.x * 4;
  u32x2 v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = *(const u32x2 G1*)(in + base + 1024 * k);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const f32x4 b = *(const f32x4 G1*)(bucket + ((threadIdx.x * 4 + 64 * k) & 1023));
    f32x4 o = {(float)(short)(v[k][0] & 0xffff) * b[0], (float)(short)(v[k][0] >> 16) * b[1],
               (float)(short)(v[k][1] & 0xffff) * b[2], (float)(short)(v[k][1] >> 16) * b[3]};
    *(f32x4 G1*)(out + base + 1024 * k) = o;
  }
}
// s9: s1 behind two dependent descriptor loads (chunk -> tensor index -> pointers), as the grouped kernels
struct Desc { const short* in; float* out; long long pad[6]; };
__global__ __launch_bounds__(256) void s9(const Desc* ds, const int* cmap) {
  const Desc* d = &ds[cmap[blockIdx.x]];
  const short* in = d->in; float* out = d->out;
  const long long base = (long long)blockIdx.x * CH + threadIdx.x * 4;
  u32x2 v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = *(const u32x2 G1*)(in + base + 1024 * k);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    f32x4 o = {(float)(short)(v[k][0] & 0xffff), (float)(short)(v[k][0] >> 16), (float)(short)(v[k][1] & 0xffff),
               (float)(short)(v[k][1] >> 16)};
    *(f32x4 G1*)(out + base + 1024 * k) = o;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  short* codes; float *fin, *fout;
  CK(hipMalloc(&codes, NEL * 2)); CK(hipMalloc(&fin, NEL * 4)); CK(hipMalloc(&fout, NEL * 4));
  CK(hipMemset(codes, 1, NEL * 2)); CK(hipMemset(fin, 0, NEL * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned grid = (unsigned)(NEL / CH);
  auto time = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 200; ++i) launch();   // ~0.1 s of warm-up: clocks
    (void)hipEventRecord(e0, 0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-4s %8.1f us  %6.2f TB/s\n", name, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
  };
  time("s1", [&] { hipLaunchKernelGGL(s1, dim3(grid), dim3(256), 0, 0, codes, fout); }, NEL * 6.0);
  time("s2", [&] { hipLaunchKernelGGL(s2, dim3(grid), dim3(256), 0, 0, codes, fout); }, NEL * 6.0);
  time("s6", [&] { hipLaunchKernelGGL(s6, dim3(grid), dim3(256), 0, 0, codes, fout); }, NEL * 6.0);
  float* bucket; CK(hipMalloc(&bucket, 4096)); CK(hipMemset(bucket, 0, 4096));
  Desc hd; hd.in = codes; hd.out = fout;
  Desc* dd; CK(hipMalloc(&dd, sizeof(Desc) * 512));
  for (int i = 0; i < 512; ++i) CK(hipMemcpy(dd + i, &hd, sizeof(Desc), hipMemcpyHostToDevice));
  int* cmap; CK(hipMalloc(&cmap, sizeof(int) * grid));
  { int* h = (int*)malloc(sizeof(int) * grid); for (unsigned i = 0; i < grid; ++i) h[i] = i % 395;
    CK(hipMemcpy(cmap, h, sizeof(int) * grid, hipMemcpyHostToDevice)); free(h); }
  time("s7", [&] { hipLaunchKernelGGL(s7, dim3(grid), dim3(256), 0, 0, codes, fout, bucket); }, NEL * 6.0);
  time("s8", [&] { hipLaunchKernelGGL(s8, dim3(grid), dim3(256), 0, 0, codes, fout, bucket); }, NEL * 6.0);
  time("s9", [&] { hipLaunchKernelGGL(s9, dim3(grid), dim3(256), 0, 0, dd, cmap); }, NEL * 6.0);
  time("s3", [&] { hipLaunchKernelGGL(s3, dim3(grid), dim3(256), 0, 0, fin, fout); }, NEL * 8.0);
  time("s4", [&] { hipLaunchKernelGGL(s4, dim3(grid), dim3(256), 0, 0, fin, fout); }, NEL * 4.0);
  time("s5", [&] { hipLaunchKernelGGL(s5, dim3(grid), dim3(256), 0, 0, fin, fout); }, NEL * 4.0);
  return 0;
}
