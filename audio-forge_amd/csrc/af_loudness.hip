// af_loudness.hip -- K-weighted 100 ms energy sums for the offline integrated-loudness operator
// (`measure_integrated_loudness`, rust-core/src/lib.rs:290-298 over dsp/loudness.rs:43-83: ebur128
// `Mode::I | Mode::HISTOGRAM`, mono).  The K-weighting filter is a 4th-order recurrence per stream, so the
// kernel is lane-per-stream over 64 x 64 LDS tiles; it emits sum(y^2) per 100 ms, from which the host forms
// the 400 ms gating blocks and the histogram gate (a few hundred blocks per stream -- not worth a kernel).
#include <hip/hip_runtime.h>

#include "af_device.h"

namespace af {

struct LoudnessArgs {
  const float *audio;     // [stream][stride]
  double *partial;        // [stream][n100]
  int32_t *non_finite;    // [stream] set to 1 when a sample is NaN / Inf
  double b[5], a[5];      // K-weighting (ebur128 filter, direct form II)
  int64_t n_samples, stride, n100;
  int32_t n_streams, s100;
};

__global__ __launch_bounds__(kLanes) void kweight_energy_kernel(LoudnessArgs a) {
  __shared__ float x[kTile][kLanes + 1];
  const int lane = threadIdx.x;
  const int s0 = blockIdx.x * kLanes;
  const int s = s0 + lane;
  const bool valid = s < a.n_streams;
  double v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, acc = 0.0;
  int in_block = 0;
  int64_t block = 0;
  int bad = 0;
  const int64_t n_used = a.n100 * a.s100;  // trailing samples short of 100 ms never reach a gating block
  for (int64_t t0 = 0; t0 < a.n_samples; t0 += kTile) {
    const int len = (int)((a.n_samples - t0) < kTile ? (a.n_samples - t0) : kTile);
    for (int r = 0; r < kLanes; ++r) {
      const int sr = s0 + r;
      float v = 0.0f;
      if (sr < a.n_streams && lane < len) v = a.audio[(int64_t)sr * a.stride + t0 + lane];
      x[lane][r] = v;
    }
    __syncthreads();
    for (int t = 0; t < len; ++t) {
      const float xin = x[t][lane];
      if ((__float_as_uint(xin) & 0x7f800000u) == 0x7f800000u) bad = 1;
      if (t0 + t < n_used) {
        const double v0 = (double)xin - a.a[1] * v1 - a.a[2] * v2 - a.a[3] * v3 - a.a[4] * v4;
        const double y = a.b[0] * v0 + a.b[1] * v1 + a.b[2] * v2 + a.b[3] * v3 + a.b[4] * v4;
        v4 = v3; v3 = v2; v2 = v1; v1 = v0;
        acc += y * y;
        if (++in_block == a.s100) {
          if (valid) a.partial[(int64_t)s * a.n100 + block] = acc;
          acc = 0.0;
          in_block = 0;
          ++block;
        }
      }
    }
    __syncthreads();
  }
  if (valid) a.non_finite[s] = bad;
}

hipError_t launch_kweight_energy(const float *audio, double *partial, int32_t *non_finite, const double b[5],
                                 const double a5[5], int64_t n_samples, int64_t stride, int64_t n100, int32_t n_streams,
                                 int32_t s100, hipStream_t stream) {
  LoudnessArgs a{};
  a.audio = audio;
  a.partial = partial;
  a.non_finite = non_finite;
  for (int i = 0; i < 5; ++i) { a.b[i] = b[i]; a.a[i] = a5[i]; }
  a.n_samples = n_samples; a.stride = stride; a.n100 = n100; a.n_streams = n_streams; a.s100 = s100;
  hipLaunchKernelGGL(kweight_energy_kernel, dim3((n_streams + kLanes - 1) / kLanes), dim3(kLanes), 0, stream, a);
  return hipGetLastError();
}

}  // namespace af
