// Dependent-issue latency of the vector instructions the serial stages are made of, one wave on one SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/valu_latency.hip -o tools/probe/valu_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))

template <int kKind>
__global__ __launch_bounds__(64) void chain(double *out, long long *cycles, double a, double b, float fa) {
  double x = a + threadIdx.x * 1e-9, y = b;
  float f = fa + threadIdx.x * 1e-6f, g = 0.5f;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 16; ++it) {
    if (kKind == 0) { REP256(asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));) }
    if (kKind == 1) { REP256(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));) }
    if (kKind == 2) { REP256(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y));) }
    if (kKind == 3) { REP256(asm volatile("v_cvt_f32_f64 %0, %1\n v_cvt_f64_f32 %1, %0" : "+v"(f), "+v"(x));) }
    if (kKind == 4) { REP256(asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(g));) }
    if (kKind == 5) { REP256(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f));) }
    if (kKind == 6) { REP256(asm volatile("v_cmp_gt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(x), "+v"(y), "+v"(f) : "v"(g) : "vcc");) }
    if (kKind == 7) { REP256(asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(y));) }
    if (kKind == 8) {  // two independent chains interleaved
      REP256(asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(x), "+v"(y) : "v"(a));) }
    if (kKind == 9) {  // four independent f64 adds
      double z = a * 3, u = b * 5;
      REP256(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(x), "+v"(y), "+v"(z), "+v"(u) : "v"(a));)
      x += z + u;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[threadIdx.x] = x + y + f + g;
  if (threadIdx.x == 0) {
    cycles[0] = t1 - t0;                 // s_memtime: shader cycles (MI355X_MICROARCH.md, "s_memtime tick = shader cycle")
    cycles[1] = (long long)(r1 - r0);    // s_memrealtime: the constant 100 MHz counter
  }
}

template <int kKind>
void run(const char *name, int per_rep) {
  double *out; long long *cyc;
  hipMalloc(&out, 64 * sizeof(double));
  hipMalloc(&cyc, 2 * sizeof(long long));
  hipLaunchKernelGGL(chain<kKind>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0, 1.0000001, 1.0f);
  hipLaunchKernelGGL(chain<kKind>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0, 1.0000001, 1.0f);
  long long h[2] = {0, 0};
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  // readcyclecounter = s_memtime = shader cycles; the in-kernel clock is d(s_memtime) / d(s_memrealtime) x 100 MHz
  const int n = 16 * 256 * per_rep;
  const double ghz = h[1] > 0 ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
  printf("%-34s %8lld cycles for %d instructions -> %.2f cycles each = %.2f ns at the in-kernel clock of %.2f GHz\n", name, h[0], n,
         (double)h[0] / n, ghz > 0 ? (double)h[0] / n / ghz : 0.0, ghz);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("v_add_f64 dependent", 1);
  run<1>("v_mul_f64 dependent", 1);
  run<2>("v_fma_f64 dependent", 1);
  run<3>("cvt f64->f32->f64 pair", 2);
  run<4>("v_add_f32 dependent", 1);
  run<5>("v_mov_b32_dpp row_shr:1 dependent", 1);
  run<6>("v_cmp_gt_f64 + v_cndmask", 2);
  run<7>("v_max_f64 dependent", 1);
  run<8>("2 independent v_add_f64 chains", 2);
  run<9>("4 independent v_add_f64 chains", 4);
  return 0;
}
