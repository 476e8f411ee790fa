// Dev (round 5): how many dependent tiny dispatches per second one stream / several streams sustain
// (the tridiagonalisation issues three dependent launches per column and stream group).
// build: hipcc -O3 --offload-arch=gfx950 tools/bench_dispatch.hip -o tools/bin/bench_dispatch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

__global__ void tiny(float* p, int wgs) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void busy(float* p, int iters) {   // ~iters * 4 cycles of dependent work per thread
  float x = p[blockIdx.x & 1023];
  for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
  if (x == 123.f) p[0] = x;
}

int main() {
  float* d; hipMalloc(&d, 4096 * 4); hipMemset(d, 0, 4096 * 4);
  hipStream_t s[8];
  for (int i = 0; i < 8; ++i) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
  const int N = 24000;
  for (int ns : {1, 2, 4, 8}) {
    for (int grid : {1, 64, 2048}) {
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s[i % ns], d + 64 * (i % ns), grid);
      auto t1 = std::chrono::steady_clock::now();
      hipDeviceSynchronize();
      auto t2 = std::chrono::steady_clock::now();
      const double host = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
      const double all = std::chrono::duration<double, std::micro>(t2 - t0).count() / N;
      printf("streams %d grid %4d: host enqueue %.2f us/launch, end to end %.2f us/launch\n", ns, grid, host, all);
    }
  }
  // a 20 us kernel (fills the chip) on stream 0 alternating with tiny kernels on the others: do the tiny ones hide?
  for (int ns : {1, 2, 4}) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    const int M = 4000;
    for (int i = 0; i < M; ++i)
      for (int g = 0; g < ns; ++g) {
        hipLaunchKernelGGL(busy, dim3(2048 / ns), dim3(256), 0, s[g], d, 3000);
        hipLaunchKernelGGL(tiny, dim3(16), dim3(128), 0, s[g], d + 64 * g, 1);
        hipLaunchKernelGGL(tiny, dim3(16), dim3(128), 0, s[g], d + 64 * g, 1);
      }
    hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    printf("busy(2048/%d wgs)+2 tiny per group, %d groups: %.2f us per round\n", ns, ns,
           std::chrono::duration<double, std::micro>(t2 - t0).count() / M);
  }
  return 0;
}
