// af_resampler.hip -- batched asynchronous sinc resampler for gfx950 (the product resampler of
// rust-core/src/audio/processor/resampling.rs:140-261 over rubato's SincFixedIn, see af_resampler_host.hpp).
//
// Work decomposition.  Output positions are the same for every stream, so a workgroup takes 64 streams
// (lane = stream) x one segment of consecutive outputs:
//   1. the input span the segment reads (segment * 1/ratio + sinc_len + 3 frames) moves HBM -> LDS once,
//      transposed to [time][stream] (each stream row is read as coalesced 512 B pieces; 65-double rows keep
//      the transposing writes at the natural 2-way bank split of 8-byte accesses);
//   2. each wave takes a run of outputs, two at a time; per output the four sinc rows are WAVE-UNIFORM (scalar
//      loads, SGPR operands), the signal is one conflict-free ds_read_b64 per tap shared by the eight rows of
//      the pair, and the arithmetic is 4 x sinc_len f64 FMAs per output per lane -- the kernel is bound by the
//      f64 VALU rate (512 FMA per output frame at sinc_len 128), not by HBM: 16 B of audio per 1 kFLOP.
//      (One output per pass read LDS once per 4 FMAs, which saturates the LDS pipe exactly when the VALU
//      saturates; pairing halves that.)
//   3. results go back through LDS so that the store is again coalesced along time.
// Every sinc row is stored with eight zero taps either side, so the window offsets of the four points (0..2
// frames) and of the pair's second output (its window starts 0..6 frames later) need no branches: a zero tap
// leaves the accumulator unchanged, and each accumulator is one fused multiply-add chain over the taps in
// increasing order -- the oracle's order, so the two agree bit for bit.
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

// rubato interp_cubic: the cubic through the four points, evaluated at `f` between y1 and y2
__device__ __forceinline__ double interp_cubic(double f, double y0, double y1, double y2, double y3) {
  const double a1 = -(1.0 / 3.0) * y0 - 0.5 * y1 + y2 - (1.0 / 6.0) * y3;
  const double a2 = 0.5 * (y0 + y2) - y1;
  const double a3 = 0.5 * (y1 - y2) + (1.0 / 6.0) * (y3 - y0);
  const double f2 = f * f;
  const double f3 = f2 * f;
  return y1 + a1 * f + a2 * f2 + a3 * f3;
}

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

  // ---- 2. the outputs of this wave, in pairs
  double res[kOutPerWave];
#pragma unroll
  for (int j = 0; j < kOutPerWave; j += 2) {
    const int oa = wave * kOutPerWave + j;
    res[j] = 0.0;
    res[j + 1] = 0.0;
    if (oa < n_seg) {
      const bool has_b = oa + 1 < n_seg;
      const ResamplePos pa = a.pos[o0 + oa];                  // wave-uniform
      const ResamplePos pb = a.pos[o0 + oa + (has_b ? 1 : 0)];
      const int row0 = (int)(pa.base - first);
      const int delta = (int)(pb.base - pa.base);             // 0 .. 6 (host-checked)
      const double *__restrict__ ca0 = a.table + (int)pa.sub[0] * stride + (kResampleTablePad - (int)pa.off[0]);
      const double *__restrict__ ca1 = a.table + (int)pa.sub[1] * stride + (kResampleTablePad - (int)pa.off[1]);
      const double *__restrict__ ca2 = a.table + (int)pa.sub[2] * stride + (kResampleTablePad - (int)pa.off[2]);
      const double *__restrict__ ca3 = a.table + (int)pa.sub[3] * stride + (kResampleTablePad - (int)pa.off[3]);
      const double *__restrict__ cb0 = a.table + (int)pb.sub[0] * stride + (kResampleTablePad - (int)pb.off[0] - delta);
      const double *__restrict__ cb1 = a.table + (int)pb.sub[1] * stride + (kResampleTablePad - (int)pb.off[1] - delta);
      const double *__restrict__ cb2 = a.table + (int)pb.sub[2] * stride + (kResampleTablePad - (int)pb.off[2] - delta);
      const double *__restrict__ cb3 = a.table + (int)pb.sub[3] * stride + (kResampleTablePad - (int)pb.off[3] - delta);
      const double *x = &lds[row0 * kResRowStride + lane];
      double ya0 = 0.0, ya1 = 0.0, ya2 = 0.0, ya3 = 0.0, yb0 = 0.0, yb1 = 0.0, yb2 = 0.0, yb3 = 0.0;
      const int taps = L + 2 + delta;
#pragma unroll 4
      for (int k = 0; k < taps; ++k) {
        const double v = x[k * kResRowStride];
        ya0 = __builtin_fma(v, ca0[k], ya0);
        ya1 = __builtin_fma(v, ca1[k], ya1);
        ya2 = __builtin_fma(v, ca2[k], ya2);
        ya3 = __builtin_fma(v, ca3[k], ya3);
        yb0 = __builtin_fma(v, cb0[k], yb0);
        yb1 = __builtin_fma(v, cb1[k], yb1);
        yb2 = __builtin_fma(v, cb2[k], yb2);
        yb3 = __builtin_fma(v, cb3[k], yb3);
      }
      res[j] = interp_cubic(pa.frac, ya0, ya1, ya2, ya3);
      res[j + 1] = interp_cubic(pb.frac, yb0, yb1, yb2, yb3);
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

// ---------------------------------------------------------------------------------------------------------
// Matrix-core variant (default for ratios near 1): the same table-driven FIR as a small GEMM per tile.
//   D[stream][(output, phase)] += X[stream][tap] * C[tap][(output, phase)]      v_mfma_f64_16x16x4_f64
// 16 streams x (4 outputs x 4 phases) per tile, taps in steps of 4.  Probed on gfx950 (tools/probe/
// mfma_f64_probe.hip): the instruction is an exact k-ascending fused multiply-add chain starting from C, with
// A[i = l % 16][k = l / 16], B[k = l / 16][j = l % 16], D[i = l / 16 + 4 r][j = l % 16] -- so chaining it over the tap
// blocks reproduces the oracle's single fma chain per (stream, row) bit for bit, zero pad taps included.
// Why it is faster than the VALU form although the f64 matrix and vector peaks are equal on MI355X: operands.
// The signal is ONE conflict-free ds_read_b64 per 1024 FMAs (LDS tile [stream group][time][16 streams]), the
// coefficients ONE per-lane 8-byte load per 4096 FMAs (all four stream groups reuse it) and can be fetched far
// ahead, so neither the LDS pipe nor the scalar cache sits next to the arithmetic any more.
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kMfSeg = 128, kMfRows = 288;

template <int J>
__device__ __forceinline__ double quad_bcast(double v) {
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffll), J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// kGroups stream groups of 16 per workgroup.  Four groups (64 streams, 16 waves, one 147 KB workgroup per CU) is the
// default.  Two groups (32 streams, 8 waves, 74 KB) let two workgroups share a CU so that one computes while the
// other moves its tile -- measured 70 ms against 53 ms for the same job: halving the reuse of every coefficient
// load costs more than the overlap wins (AF_RESAMPLER_VARIANT=mfma32 keeps it selectable).
template <int kGroups>
__global__ __launch_bounds__(kGroups * 4 * kResLanes) void resample_mfma_kernel(ResampleArgs a) {
  constexpr int kWaves = kGroups * 4;             // 32 tiles of four outputs over the waves
  constexpr int kTilesPerWave = 32 / kWaves;
  constexpr int kStreams = 16 * kGroups;
  constexpr int kOutStride = kStreams + 1;
  extern __shared__ double lds[];  // [kGroups][kMfRows][16 streams]; later [128 outputs][kStreams + 1]
  const int tid = threadIdx.x;
  const int lane = tid & (kResLanes - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kResLanes);
  const int m = lane & 15, kq = lane >> 4;
  const int64_t o0 = (int64_t)blockIdx.x * kMfSeg;
  const int n_seg = (int)((a.n_out - o0) < kMfSeg ? (a.n_out - o0) : kMfSeg);
  const int s0 = blockIdx.y * kStreams;
  const int L = a.sinc_len;
  const int stride = L + 2 * kResampleTablePad;
  const int64_t first = a.pos[o0].base;
  const int64_t last = a.pos[o0 + n_seg - 1].base + 2 + L - 1;
  int rows = (int)(last - first + 1) + 8;  // + the tap-block round-up of the last tile (zero taps: values only need to be finite)
  if (rows > kMfRows) rows = kMfRows;

  // ---- 1. input span -> LDS; lane = (stream m of the group, time offset kq): 512 contiguous LDS bytes per instruction.
  // All of a wave's loads are issued before the first one is consumed (a load-wait-store loop would pay the HBM
  // latency once per row block).
  {
    const int g = wave % kGroups;
    const int s = s0 + 16 * g + m;
    const double *src = a.in + (int64_t)s * a.in_stride;
    constexpr int kBlocks = (kMfRows / 4 + 3) / 4;  // row blocks of 4 per wave (four waves share a stream group)
    double v[kBlocks];
#pragma unroll
    for (int i = 0; i < kBlocks; ++i) {
      const int t = ((wave / kGroups) + 4 * i) * 4 + kq;
      const int64_t gi = first + t;
      v[i] = (s < a.n_streams && t < rows && gi >= 0 && gi < a.n_in) ? src[gi] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < kBlocks; ++i) {
      const int t = ((wave / kGroups) + 4 * i) * 4 + kq;
      if (t < rows) lds[(g * kMfRows + t) * 16 + m] = v[i];
    }
  }
  __syncthreads();

  // ---- 2. tiles of four outputs
  const int oc = m >> 2, ph = m & 3;  // this lane's column: output oc of the tile, phase ph
  double res[kTilesPerWave][kGroups][4];
#pragma unroll
  for (int gi = 0; gi < kTilesPerWave; ++gi) {
    const int ob = (wave * kTilesPerWave + gi) * 4;
#pragma unroll
    for (int g = 0; g < kGroups; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) res[gi][g][r] = 0.0;
    if (ob < n_seg) {
      const int o_lane = (ob + oc) < n_seg ? (ob + oc) : (n_seg - 1);
      const int o_last = (ob + 3) < n_seg ? (ob + 3) : (n_seg - 1);
      const ResamplePos P = a.pos[o0 + o_lane];
      const int64_t base0 = a.pos[o0 + ob].base;
      const int row0 = (int)(base0 - first);
      const int delta = (int)(P.base - base0);
      const int delta_max = (int)(a.pos[o0 + o_last].base - base0);
      const double *__restrict__ bp = a.table + (int)P.sub[ph] * stride + (kResampleTablePad - (int)P.off[ph] - delta) + kq;
      const int ksteps = (L + 2 + delta_max + 3) >> 2;
      f64x4 acc[kGroups];
#pragma unroll
      for (int g = 0; g < kGroups; ++g) acc[g] = f64x4{0, 0, 0, 0};
      const double *xa = &lds[(row0 + kq) * 16 + m];
      // coefficients are fetched four tap blocks ahead of the MFMAs that use them
      double bq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) bq[u] = bp[4 * (u < ksteps ? u : ksteps - 1)];
      for (int kk = 0; kk < ksteps; kk += 4) {
        double bn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int nk = kk + 4 + u;
          bn[u] = bp[4 * (nk < ksteps ? nk : ksteps - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (kk + u < ksteps) {
            const double *xr = xa + (kk + u) * 64;
#pragma unroll
            for (int g = 0; g < kGroups; ++g)
              acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[g * kMfRows * 16], bq[u], acc[g], 0, 0, 0);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) bq[u] = bn[u];
      }
      // the four phases of one (stream, output) sit in the four lanes of a quad: exchange, then the cubic
      const double f = P.frac;
#pragma unroll
      for (int g = 0; g < kGroups; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = acc[g][r];
          res[gi][g][r] = interp_cubic(f, quad_bcast<0>(v), quad_bcast<1>(v), quad_bcast<2>(v), quad_bcast<3>(v));
        }
    }
  }
  __syncthreads();

  // ---- 3. transposed store through LDS ([output][streams + 1]); D row i = kq + 4 r of stream group g
#pragma unroll
  for (int gi = 0; gi < kTilesPerWave; ++gi) {
    const int o = (wave * kTilesPerWave + gi) * 4 + oc;
    if (ph == 0) {
#pragma unroll
      for (int g = 0; g < kGroups; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[o * kOutStride + 16 * g + kq + 4 * r] = res[gi][g][r];
    }
  }
  __syncthreads();
  for (int r = wave; r < kStreams; r += kWaves) {
    const int s = s0 + r;
    if (s >= a.n_streams) continue;
    double *dst = a.out + (int64_t)s * a.out_stride + o0;
    for (int t = lane; t < n_seg; t += kResLanes) dst[t] = lds[t * kOutStride + r];
  }
}

template <int kGroups>
static hipError_t launch_resample_mfma_variant(const ResampleArgs &a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(resample_mfma_kernel<kGroups>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) return err;
    attr_set = true;
  }
  const size_t dyn = sizeof(double) * kGroups * kMfRows * 16;
  const dim3 grid((unsigned)((a.n_out + kMfSeg - 1) / kMfSeg), (unsigned)((a.n_streams + 16 * kGroups - 1) / (16 * kGroups)));
  hipLaunchKernelGGL(resample_mfma_kernel<kGroups>, grid, dim3(kGroups * 4 * kResLanes), dyn, stream, a);
  return hipGetLastError();
}
static hipError_t launch_resample_mfma(const ResampleArgs &a, int variant, hipStream_t stream) {
  return variant == 2 ? launch_resample_mfma_variant<2>(a, stream) : launch_resample_mfma_variant<4>(a, stream);
}

// the matrix-core tile needs: 128 outputs' span + the tap round-up inside 288 rows, and a tile's four windows
// starting within 8 frames of each other (row padding 16)
bool resample_mfma_ok(double ratio, int sinc_len) {
  return std::ceil(128.0 / ratio) + sinc_len + 14 <= kMfRows && 3.0 / ratio + 3.0 <= 8.0;  // (128 x 33 staging fits 2 x 288 x 16)
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
  if (1.0 / ratio > 5.0) return 0;  // a pair's second window may start at most 6 frames after the first (row padding)
  const int candidates[] = {128, 64, 32, 16, 8};
  for (int seg : candidates) {
    const double span = std::ceil((double)seg / ratio) + sinc_len + 4;
    if (span <= kResMaxRows && seg <= kResMaxRows) return seg;
  }
  return 0;
}

hipError_t launch_resample(const double *in, double *out, const ResamplePos *pos, const double *table, int64_t n_in,
                           int64_t n_out, int64_t in_stride, int64_t out_stride, int32_t n_streams, int32_t sinc_len,
                           double ratio, int variant, hipStream_t stream) {
  ResampleArgs a{in, out, pos, table, n_in, n_out, in_stride, out_stride, n_streams, sinc_len, kResMaxRows};
  if (n_out <= 0 || n_streams <= 0) return hipSuccess;
  if (variant != 1 && resample_mfma_ok(ratio, sinc_len)) return launch_resample_mfma(a, variant, stream);
  switch (resample_segment_outputs(ratio, sinc_len)) {
    case 128: return launch_resample_variant<16, 8>(a, stream);
    case 64: return launch_resample_variant<16, 4>(a, stream);
    case 32: return launch_resample_variant<16, 2>(a, stream);
    case 16: return launch_resample_variant<8, 2>(a, stream);
    case 8: return launch_resample_variant<4, 2>(a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace af
