// diag.hip — diagnostic entry points (never on the product path):
//   ps_diag_mfma_clock : the shader clock the chip holds under fp32-MFMA load, measured live
//                        (delta s_memtime / delta s_memrealtime x 100 MHz around an MFMA loop on
//                        non-trivial operands, MI355X_MICROARCH.md "DVFS give-back" item 6) and
//                        the fp32 MFMA rate that loop reaches.  bench.py prints both next to
//                        every roofline fraction, so that a box-to-box move of a kernel's
//                        TFLOP/s can be attributed to the clock or to the kernel.
//   ps_diag_spin       : a filler kernel that holds workgroup slots for a given time (tests:
//                        the resident power iteration next to a kernel on another stream).
//   ps_collective_in_flight / ps_power_iteration_health / _reset_health : the process-wide
//                        health record of the resident power iteration (power_iter.hip.h).
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "power_iter.hip.h"

using namespace psk;

namespace {

typedef unsigned long long u64;

__global__ __launch_bounds__(256, 2) void diag_mfma_kernel(u64* stamps, int iters, float* sink) {
  const int tid = threadIdx.x;
  // non-trivial operands (zeros let the chip hold a higher clock than real data does)
  unsigned h = (unsigned)(blockIdx.x * 256 + tid) * 2654435761u + 12345u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  float a = ((float)(h & 0xffffu) - 32768.f) * (1.f / 33554432.f);
  float b = ((float)(h >> 16) - 32768.f) * (1.f / 33554432.f);
  f32x16 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  __syncthreads();
  const u64 c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
      a = -a;
    }
  }
  const u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {
    stamps[2 * blockIdx.x + 0] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[k][r];
  if (t == 123456.789f) sink[0] = t;   // keeps the loop alive
}

// Issue-mix microbenchmark: 16 fp32 MFMAs per step with NV independent 32-bit VALU adds (or, WIDE,
// 64-bit v_lshl_add_u64) and NS scalar adds spread between them.  Answers whether ordinary VALU
// work in a K loop takes cycles away from the fp32 MFMA pipe (the fp32 MFMA rate equals the
// packed-fp32 VALU rate on this part).
template <int NV, bool WIDE>
__global__ __launch_bounds__(256, 2) void diag_mix_kernel(int iters, float* sink) {
  const int tid = threadIdx.x;
  unsigned h = (unsigned)(blockIdx.x * 256 + tid) * 2654435761u + 12345u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  float a = ((float)(h & 0xffffu) - 32768.f) * (1.f / 33554432.f);
  float b = ((float)(h >> 16) - 32768.f) * (1.f / 33554432.f);
  f32x16 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  unsigned x[4] = {h, h + 1, h + 2, h + 3};
  unsigned long long w[4] = {h, h + 1ull, h + 2ull, h + 3ull};
  const unsigned y = h | 1u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV / 16; ++v) {
          if (WIDE) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[(k + v) & 3]) : "v"(w[(k + v + 1) & 3]));
          else asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[(k + v) & 3]) : "v"(y));
        }
        if ((NV % 16) != 0 && (u * 4 + k) < (NV % 16)) {
          if (WIDE) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[k & 3]) : "v"(w[(k + 1) & 3]));
          else asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k & 3]) : "v"(y));
        }
      }
      a = -a;
    }
  }
  float t = (float)(x[0] + x[1] + x[2] + x[3]) + (float)(w[0] + w[1] + w[2] + w[3]);
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[k][r];
  if (t == 123456.789f) sink[0] = t;
}

__global__ void diag_spin_kernel(u64 ticks) {
  extern __shared__ float spin_lds[];
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) spin_lds[0] = 0.f;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

}  // namespace

// Runs the MFMA loop back to back for about `warm_ms` (so that the chip settles at the clock
// it holds under this load), then measures one launch.  Synchronises the stream; allocates
// and frees 2 small device buffers (diagnostic call, not a compute call).
extern "C" int ps_diag_mfma_clock(void* stream, double warm_ms, double* clock_ghz,
                                  double* mfma_f32_tflops) {
  PS_DEVICE_CHECK();
  hipStream_t st = (hipStream_t)stream;
  int dev = 0;
  hipDeviceProp_t prop;
  PS_HIP(hipGetDevice(&dev));
  PS_HIP(hipGetDeviceProperties(&prop, dev));
  const int grid = 2 * prop.multiProcessorCount;   // 2 workgroups per CU = 2 waves per SIMD
  const int iters = 4096;                          // x 16 MFMAs of 64 cycles: ~1.8 ms at 2.3 GHz
  u64* stamps = nullptr;
  float* sink = nullptr;
  PS_HIP(hipMalloc((void**)&stamps, sizeof(u64) * 2 * grid));
  if (hipMalloc((void**)&sink, 64) != hipSuccess) { (void)hipFree(stamps); return PS_EINTERNAL; }
  hipEvent_t e0, e1;
  int rc = 0;
  if ((rc = (int)hipEventCreate(&e0)) || (rc = (int)hipEventCreate(&e1))) {
    (void)hipFree(stamps); (void)hipFree(sink); return rc;
  }
  const int warm_launches = std::max(1, (int)(warm_ms / 1.8));
  for (int i = 0; i < warm_launches; ++i)
    hipLaunchKernelGGL(diag_mfma_kernel, dim3(grid), dim3(256), 0, st, stamps, iters, sink);
  (void)hipEventRecord(e0, st);
  hipLaunchKernelGGL(diag_mfma_kernel, dim3(grid), dim3(256), 0, st, stamps, iters, sink);
  (void)hipEventRecord(e1, st);
  rc = (int)hipStreamSynchronize(st);
  float ms = 0.f;
  if (!rc) rc = (int)hipEventElapsedTime(&ms, e0, e1);
  std::vector<u64> h(2 * grid);
  if (!rc) rc = (int)hipMemcpy(h.data(), stamps, sizeof(u64) * 2 * grid, hipMemcpyDeviceToHost);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(stamps); (void)hipFree(sink);
  if (rc) return rc;
  std::vector<double> ghz(grid);
  for (int i = 0; i < grid; ++i)
    ghz[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 0.1 : 0.0;   // x 100 MHz
  std::sort(ghz.begin(), ghz.end());
  if (clock_ghz) *clock_ghz = ghz[grid / 2];
  if (mfma_f32_tflops)
    *mfma_f32_tflops = (double)grid * 4.0 * iters * 16.0 * 4096.0 / (ms * 1e-3) / 1e12;
  return PS_OK;
}

// fp32 MFMA rate (TFLOP/s) of a loop of 16 MFMAs + `valu_per_16` VALU adds (wide != 0: 64-bit
// adds), `wgs_per_cu` (1 | 2) workgroups per CU.  Diagnostic; synchronises the stream.
extern "C" int ps_diag_mfma_mix(void* stream, int valu_per_16, int wide, int wgs_per_cu,
                                double* mfma_f32_tflops) {
  PS_DEVICE_CHECK();
  hipStream_t st = (hipStream_t)stream;
  int dev = 0;
  hipDeviceProp_t prop;
  PS_HIP(hipGetDevice(&dev));
  PS_HIP(hipGetDeviceProperties(&prop, dev));
  const int grid = (wgs_per_cu == 1 ? 1 : 2) * prop.multiProcessorCount;
  const int iters = 4096;
  float* sink = nullptr;
  PS_HIP(hipMalloc((void**)&sink, 64));
  hipEvent_t e0, e1;
  int rc = 0;
  if ((rc = (int)hipEventCreate(&e0)) || (rc = (int)hipEventCreate(&e1))) { (void)hipFree(sink); return rc; }
  auto launch = [&]() {
#define PS_MIX(NV)                                                                              \
  if (valu_per_16 == NV) {                                                                      \
    if (wide) hipLaunchKernelGGL((diag_mix_kernel<NV, true>), dim3(grid), dim3(256), 0, st, iters, sink); \
    else hipLaunchKernelGGL((diag_mix_kernel<NV, false>), dim3(grid), dim3(256), 0, st, iters, sink);     \
    return true;                                                                                \
  }
    PS_MIX(0) PS_MIX(4) PS_MIX(8) PS_MIX(16) PS_MIX(32) PS_MIX(64)
#undef PS_MIX
    return false;
  };
  bool ok = true;
  for (int i = 0; i < 200 && ok; ++i) ok = launch();
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < 10 && ok; ++i) ok = launch();
  (void)hipEventRecord(e1, st);
  rc = (int)hipStreamSynchronize(st);
  float ms = 0.f;
  if (!rc) rc = (int)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  if (!ok) return PS_EINVAL;
  if (rc) return rc;
  if (mfma_f32_tflops)
    *mfma_f32_tflops = 10.0 * (double)grid * 4.0 * iters * 16.0 * 4096.0 / (ms * 1e-3) / 1e12;
  return PS_OK;
}

extern "C" int ps_diag_spin(void* stream, int workgroups, int threads, int lds_bytes, double ms) {
  PS_DEVICE_CHECK();
  if (workgroups < 1 || threads < 64 || threads > 1024 || lds_bytes < 4 || lds_bytes > 160 * 1024)
    return PS_EINVAL;
  static bool attr = false;
  if (!attr) {
    PS_HIP(hipFuncSetAttribute((const void*)diag_spin_kernel,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  hipLaunchKernelGGL(diag_spin_kernel, dim3(workgroups), dim3(threads), (size_t)lds_bytes,
                     (hipStream_t)stream, (u64)(ms * 1.0e5));
  PS_LAUNCH_CHECK();
  return PS_OK;
}

extern "C" int ps_collective_in_flight(int delta) {
  return PiPlan::health().collectives.fetch_add(delta) + delta;
}

extern "C" int ps_power_iteration_health(unsigned* expired_waits, int* collectives_in_flight,
                                         int* resident_enabled) {
  if (expired_waits) *expired_waits = PiPlan::expired_total();
  if (collectives_in_flight) *collectives_in_flight = PiPlan::health().collectives.load();
  if (resident_enabled) *resident_enabled = PiPlan::resident_enabled() ? 1 : 0;
  return PS_OK;
}

extern "C" int ps_power_iteration_reset_health(void) {
  PiPlan::PiHealth& h = PiPlan::health();
  if (h.expired) *reinterpret_cast<volatile unsigned*>(h.expired) = 0;
  return PS_OK;
}
