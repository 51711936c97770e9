// Probe: latency of a dependent chain of v_mfma_f32_16x16x4_f32 (and what clock64() ticks are worth) on gfx950.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void chain(float *out, long long *ticks, int n) {
  v4f acc = {0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  const long long t0 = clock64();
  for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  const long long t1 = clock64();
  out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
  float *d; long long *t; hipMalloc(&d, 4096); hipMalloc(&t, 8 * 1024);
  const int n = 100000;
  for (int blocks : {1, 256, 1024}) {
    hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, d, t, 1000);
    hipDeviceSynchronize();
    auto w0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, d, t, n);
    hipDeviceSynchronize();
    const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - w0).count();
    long long ticks; hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost);
    printf("blocks %4d: %.1f ns per dependent MFMA (wall), %.2f clock64 ticks per MFMA, %.2f ns per tick\n", blocks, ns / n,
           (double)ticks / n, ns / (double)ticks);
  }
  return 0;
}
