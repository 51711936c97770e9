// Probe: layout and accumulation order of v_mfma_f64_16x16x4_f64 on gfx950.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off mfma_f64_probe.hip -o mfma_f64_probe && ./mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(const double *A, const double *B, const double *C, double *D) {
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + l / 16];   // A[i][k], i = l % 16, k = l / 16
  const double b = B[(l / 16) * 16 + l % 16];  // B[k][j], k = l / 16, j = l % 16
  d4 c;
  for (int r = 0; r < 4; ++r) c[r] = 0.0;  // C = 0: the layout question is about A, B and D only
  d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = d[r];  // raw: [lane][register]
}
int main() {
  std::vector<double> A(64), B(64), C(256), D(256);
  srand(3);
  auto rnd = [] { return (rand() / (double)RAND_MAX - 0.5) * std::ldexp(1.0, rand() % 40 - 20); };
  for (auto &v : A) v = rnd();
  for (auto &v : B) v = rnd();
  for (auto &v : C) v = rnd();
  double *dA, *dB, *dC, *dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
  hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
  // reference products in both orders, C = 0
  std::vector<double> up(256), down(256);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double u = 0.0, dn = 0.0;
      for (int k = 0; k < 4; ++k) u = std::fma(A[i * 4 + k], B[k * 16 + j], u);
      for (int k = 3; k >= 0; --k) dn = std::fma(A[i * 4 + k], B[k * 16 + j], dn);
      up[i * 16 + j] = u;
      down[i * 16 + j] = dn;
    }
  // where does each raw (lane, register) value sit in the result matrix?
  int found_up = 0, found_down = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const double v = D[l * 4 + r];
      int iu = -1, ju = -1, idn = -1;
      for (int e = 0; e < 256; ++e) {
        if (up[e] == v && iu < 0) { iu = e / 16; ju = e % 16; }
        if (down[e] == v && idn < 0) idn = e;
      }
      found_up += iu >= 0;
      found_down += idn >= 0;
      if (l < 20 || l % 16 == 0) printf("lane %2d reg %d -> up(i=%d, j=%d) down_idx=%d\n", l, r, iu, ju, idn);
    }
  printf("raw values found in k-ascending reference: %d / 256, k-descending: %d / 256\n", found_up, found_down);
  return 0;
}
