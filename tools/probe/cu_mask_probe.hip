// cu_mask_probe.hip -- does hipExtStreamCreateWithCUMask confine a stream's workgroups on this platform, and which
// physical CUs (XCC, SE, CU) does mask bit i select?  Build: hipcc --offload-arch=gfx950 -O2 cu_mask_probe.hip -o cu_mask_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(unsigned *out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // keep the workgroup resident for a while so that the grid spreads over every CU the queue may use
  long long t0 = clock64();
  while (clock64() - t0 < spin) {
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (hw & 0xffff) | ((xcc & 0xf) << 16);
}

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s -> %s\n", #x, hipGetErrorString(e_));                            \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

static int run(hipStream_t s, const char *label, unsigned *d, int n) {
  std::vector<unsigned> h(n);
  hipLaunchKernelGGL(where_kernel, dim3(n), dim3(64), 0, s, d, 200000);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost));
  std::set<unsigned> cus;
  std::map<unsigned, int> per_xcc;
  for (unsigned v : h) cus.insert(v & 0xfff00ffu ? (v & 0xfffff00u) : v);  // drop the wave / simd bits below
  std::set<unsigned> ids;
  for (unsigned v : h) {
    const unsigned cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 7, xcc = (v >> 16) & 0xf;
    ids.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
  }
  for (unsigned id : ids) per_xcc[id >> 12]++;
  printf("%-28s distinct CUs: %3zu  per XCC:", label, ids.size());
  for (auto &kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
  printf("\n");
  return 0;
}

int main() {
  const int n = 4096;
  unsigned *d;
  CK(hipMalloc(&d, n * sizeof(unsigned)));
  hipStream_t plain;
  CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  if (run(plain, "unmasked", d, n)) return 1;
  struct Case { const char *label; int first, count, stride; } cases[] = {
      {"bits 0..63", 0, 64, 1}, {"bits 64..255", 64, 192, 1}, {"bits 0..31", 0, 32, 1}, {"every 4th bit (64)", 0, 64, 4},
      {"bits 0..7", 0, 8, 1}, {"bits 8..15", 8, 8, 1}};
  for (const Case &c : cases) {
    uint32_t mask[8] = {0};
    for (int k = 0; k < c.count; ++k) {
      const int bit = c.first + k * c.stride;
      mask[bit >> 5] |= 1u << (bit & 31);
    }
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) {
      printf("%-28s hipExtStreamCreateWithCUMask -> %s\n", c.label, hipGetErrorString(e));
      continue;
    }
    if (run(s, c.label, d, n)) return 1;
    CK(hipStreamDestroy(s));
  }
  return 0;
}
