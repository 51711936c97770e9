// xor_butterfly_test.hip -- wave_allsum_xor (v_permlane32/16_swap + DPP) against the __shfl_xor butterfly on random inputs: every lane of 2000 waves must match bit for bit.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
template <int kCtrl>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), kCtrl, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_allsum_xor(float acc) {
  {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
    acc = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
    acc = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  acc = acc + dpp_f<0x128>(acc);  // row_ror:8
  acc = acc + dpp_f<0x124>(acc);  // row_ror:4
  acc = acc + dpp_f<0x4E>(acc);   // quad_perm:[2,3,0,1]
  acc = acc + dpp_f<0xB1>(acc);   // quad_perm:[1,0,3,2]
  return acc;
}
__global__ void k(const float* in, float* o1, float* o2) {
  float a = in[threadIdx.x];
  float b = a;
  for (int off = 32; off >= 1; off >>= 1) b = b + __shfl_xor(b, off);
  o1[threadIdx.x] = b;
  o2[threadIdx.x] = wave_allsum_xor(a);
}
int main() {
  float *in, *o1, *o2; hipMalloc(&in, 256); hipMalloc(&o1, 256); hipMalloc(&o2, 256);
  float h[64], r1[64], r2[64];
  unsigned s = 12345; int bad = 0;
  for (int t = 0; t < 2000; ++t) {
    for (int i = 0; i < 64; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 20)) * ((s & 7) == 0 ? 1e-6f : 1.0f); }
    hipMemcpy(in, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(in, o1, o2);
    hipMemcpy(r1, o1, 256, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) if (memcmp(&r1[i], &r2[i], 4) != 0) ++bad;
  }
  printf("mismatches %d\n", bad);
  return bad != 0;
}
