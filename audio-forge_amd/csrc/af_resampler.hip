// af_resampler.hip -- batched asynchronous sinc resampler for gfx950 (the product resampler of
// rust-core/src/audio/processor/resampling.rs:140-261 over rubato's SincFixedIn, see af_resampler_host.hpp).
//
// Work decomposition.  Output positions are the same for every stream, so a workgroup takes 64 streams
// (lane = stream) x one segment of consecutive outputs:
//   1. the input span the segment reads (segment * 1/ratio + sinc_len + 3 frames) moves HBM -> LDS once,
//      transposed to [time][stream] (each stream row is read as coalesced 512 B pieces; 65-double rows keep
//      the transposing writes at the natural 2-way bank split of 8-byte accesses);
//   2. each wave takes a run of outputs; per output the four sinc rows are WAVE-UNIFORM (scalar loads,
//      SGPR operands), the signal is one conflict-free ds_read_b64 per tap shared by the four rows, and the
//      arithmetic is 4 x sinc_len f64 FMAs per lane -- the kernel is bound by the f64 VALU rate
//      (512 FMA per output frame at sinc_len 128), not by HBM: 16 B of audio per 1 kFLOP;
//   3. results go back through LDS so that the store is again coalesced along time.
// Every sinc row is stored with two zero taps either side, so the three possible window offsets of the
// four points (0..2 frames) need no branches: a zero tap leaves the accumulator unchanged, and each
// accumulator is one fused multiply-add chain over the taps in increasing order -- the oracle's order,
// so the two agree bit for bit.
#include <hip/hip_runtime.h>

#include "af_resampler_host.hpp"

namespace af {

constexpr int kResLanes = 64;
constexpr int kResRowStride = kResLanes + 1;  // doubles per LDS row

struct ResampleArgs {
  const double *in;
  double *out;
  const ResamplePos *pos;   // [n_out]
  const double *table;      // [256][sinc_len + 4]
  int64_t n_in, n_out, in_stride, out_stride;
  int32_t n_streams, sinc_len, max_rows;
};

template <int kWaves, int kOutPerWave>
__global__ __launch_bounds__(kWaves *kResLanes) void resample_kernel(ResampleArgs a) {
  extern __shared__ double lds[];  // [max(rows, segment outputs)][65]
  constexpr int kSeg = kWaves * kOutPerWave;
  const int tid = threadIdx.x;
  const int lane = tid & (kResLanes - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kResLanes);
  const int64_t o0 = (int64_t)blockIdx.x * kSeg;
  const int n_seg = (int)((a.n_out - o0) < kSeg ? (a.n_out - o0) : kSeg);
  const int s0 = blockIdx.y * kResLanes;
  const int L = a.sinc_len;
  const int stride = L + 2 * kResampleTablePad;

  const int64_t first = a.pos[o0].base;
  const int64_t last = a.pos[o0 + n_seg - 1].base + 2 + L - 1;
  int rows = (int)(last - first + 1);
  if (rows > a.max_rows) rows = a.max_rows;  // cannot happen (host sizes the segment); keeps LDS in bounds

  // ---- 1. input span -> LDS, transposed
  for (int r = wave; r < kResLanes; r += kWaves) {
    const int s = s0 + r;
    const double *src = a.in + (int64_t)s * a.in_stride;
    for (int t = lane; t < rows; t += kResLanes) {
      const int64_t g = first + t;
      double v = 0.0;
      if (s < a.n_streams && g >= 0 && g < a.n_in) v = src[g];
      lds[t * kResRowStride + r] = v;
    }
  }
  __syncthreads();

  // ---- 2. the outputs of this wave
  double res[kOutPerWave];
#pragma unroll
  for (int j = 0; j < kOutPerWave; ++j) {
    const int o = wave * kOutPerWave + j;
    res[j] = 0.0;
    if (o < n_seg) {
      const ResamplePos p = a.pos[o0 + o];  // wave-uniform
      const int row0 = (int)(p.base - first);
      const double *__restrict__ c0 = a.table + (int)p.sub[0] * stride + (kResampleTablePad - (int)p.off[0]);
      const double *__restrict__ c1 = a.table + (int)p.sub[1] * stride + (kResampleTablePad - (int)p.off[1]);
      const double *__restrict__ c2 = a.table + (int)p.sub[2] * stride + (kResampleTablePad - (int)p.off[2]);
      const double *__restrict__ c3 = a.table + (int)p.sub[3] * stride + (kResampleTablePad - (int)p.off[3]);
      const double *x = &lds[row0 * kResRowStride + lane];
      double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
      const int taps = L + 2;
#pragma unroll 8
      for (int k = 0; k < taps; ++k) {
        const double v = x[k * kResRowStride];
        y0 = __builtin_fma(v, c0[k], y0);
        y1 = __builtin_fma(v, c1[k], y1);
        y2 = __builtin_fma(v, c2[k], y2);
        y3 = __builtin_fma(v, c3[k], y3);
      }
      // rubato interp_cubic: the cubic through the four points, evaluated at `frac` between y1 and y2
      const double f = p.frac;
      const double a1 = -(1.0 / 3.0) * y0 - 0.5 * y1 + y2 - (1.0 / 6.0) * y3;
      const double a2 = 0.5 * (y0 + y2) - y1;
      const double a3 = 0.5 * (y1 - y2) + (1.0 / 6.0) * (y3 - y0);
      const double f2 = f * f;
      const double f3 = f2 * f;
      res[j] = y1 + a1 * f + a2 * f2 + a3 * f3;
    }
  }
  __syncthreads();

  // ---- 3. transposed store
#pragma unroll
  for (int j = 0; j < kOutPerWave; ++j) lds[(wave * kOutPerWave + j) * kResRowStride + lane] = res[j];
  __syncthreads();
  for (int r = wave; r < kResLanes; r += kWaves) {
    const int s = s0 + r;
    if (s >= a.n_streams) continue;
    double *dst = a.out + (int64_t)s * a.out_stride + o0;
    for (int t = lane; t < n_seg; t += kResLanes) dst[t] = lds[t * kResRowStride + r];
  }
}

constexpr int kResMaxRows = 288;  // 288 x 65 x 8 B = 149 760 B of the CU's 160 KB

template <int kWaves, int kOutPerWave>
static hipError_t launch_resample_variant(const ResampleArgs &a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(resample_kernel<kWaves, kOutPerWave>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) return err;
    attr_set = true;
  }
  constexpr int kSeg = kWaves * kOutPerWave;
  const size_t dyn = sizeof(double) * kResRowStride * (size_t)kResMaxRows;
  const dim3 grid((unsigned)((a.n_out + kSeg - 1) / kSeg), (unsigned)((a.n_streams + kResLanes - 1) / kResLanes));
  hipLaunchKernelGGL((resample_kernel<kWaves, kOutPerWave>), grid, dim3(kWaves * kResLanes), dyn, stream, a);
  return hipGetLastError();
}

// Largest segment whose input span fits the LDS tile: span = ceil(segment / ratio) + sinc_len + 3 frames.
int resample_segment_outputs(double ratio, int sinc_len) {
  const int candidates[] = {128, 64, 32, 16, 8};
  for (int seg : candidates) {
    const double span = std::ceil((double)seg / ratio) + sinc_len + 4;
    if (span <= kResMaxRows && seg <= kResMaxRows) return seg;
  }
  return 0;
}

hipError_t launch_resample(const double *in, double *out, const ResamplePos *pos, const double *table, int64_t n_in,
                           int64_t n_out, int64_t in_stride, int64_t out_stride, int32_t n_streams, int32_t sinc_len,
                           double ratio, hipStream_t stream) {
  ResampleArgs a{in, out, pos, table, n_in, n_out, in_stride, out_stride, n_streams, sinc_len, kResMaxRows};
  if (n_out <= 0 || n_streams <= 0) return hipSuccess;
  switch (resample_segment_outputs(ratio, sinc_len)) {
    case 128: return launch_resample_variant<8, 16>(a, stream);
    case 64: return launch_resample_variant<8, 8>(a, stream);
    case 32: return launch_resample_variant<8, 4>(a, stream);
    case 16: return launch_resample_variant<8, 2>(a, stream);
    case 8: return launch_resample_variant<8, 1>(a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace af
