// af_truepeak.hip -- the chain's output-side true-peak detector as a kernel of its own, on the matrix cores.
//
// `OfflineDspBlockProcessor::process_block_with_stats` ends with `TruePeakDetector::process_block(output)`
// (block_processor.rs:159, true_peak.rs:208-218): a statistic of the finished audio, feed-forward, a 4 x 32-tap FIR
// per sample (true_peak.rs:173-186) and a maximum per control block.  Inside the token-ring chain kernel it costs
// ~14 % of the instructions of a kernel that is confined to 64 CUs; here it runs after the chain launch of a window
// on the suppressor's CUs, whose matrix pipes are idle except for the network kernel.
//
// The FIR as a matrix product.  A tile is 64 consecutive samples of one stream = 16 rows of 4.  For row i (samples
// n0 + 4 i + j, j = 0..3) and column (p, j) (phase p, position j):
//     peak[n0 + 4 i + j][p] = sum_k c[p][k] x[n0 + 4 i + j - k]
//                           = sum_t B[t][(p, j)] A[i][t],   A[i][t] = x[n0 + 4 i + 3 - t],  B[t][(p, j)] = c[p][j - 3 + t]
// with t = 0..35 (B is zero where j - 3 + t falls outside 0..31).  Ascending t is ascending k for every column, and
// v_mfma_f32_16x16x4_f32 accumulates its four k in order with one fused multiply-add each (probed for the network
// kernel), so nine chained instructions from a zero accumulator give every sum as the reference's own
// `acc = mul_add(c[k], h[k], acc)`, k = 0..31, bit for bit; the zero entries leave an accumulator unchanged.
// 9 matrix instructions produce 64 samples x 4 phases: 4.5 pipe cycles per sample against 8 issue cycles per sample
// for the 128 scalar fmas.
#include <hip/hip_runtime.h>

#include "af_device.h"
#include "af_dsp.h"
#include "tp_fir_table.h"

namespace af {

typedef float tp_v4f __attribute__((ext_vector_type(4)));

struct TpDetectTable {
  float b[9][64];  // B operand of k-step s for lane: B[4 s + (lane >> 4)][lane & 15]
};

struct TpDetectArgs {
  const float *audio;     // chain output of the segment, [stream][stride] (stream-major)
  float *st32;            // state plane: rows kTpOutHist .. +31 hold the 32 samples before the segment
  BlockStats *stats;      // [block][stream] rows of the segment: output_true_peak is written here
  int64_t stream_stride;
  int64_t n_samples;      // samples per stream in the segment
  int32_t n_streams;
  int32_t control_block;
};

constexpr int kTpWavesPerGroup = 4;
constexpr int kTpMaxBlock = 1024;  // samples of one control block a wave keeps in LDS (longer blocks: chain-side detector)

__global__ __launch_bounds__(64 * kTpWavesPerGroup) void tp_detect_kernel(TpDetectArgs a, TpDetectTable tb) {
  __shared__ float xs_all[kTpWavesPerGroup][kTpTaps + kTpMaxBlock + 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_blocks = (a.n_samples + a.control_block - 1) / a.control_block;
  const int64_t unit = (int64_t)blockIdx.x * kTpWavesPerGroup + wave;  // (block, stream), streams fastest
  if (unit >= n_blocks * a.n_streams) return;  // no workgroup barrier below: waves are independent
  const int64_t b = unit / a.n_streams;
  const int s = (int)(unit - b * a.n_streams);
  const int64_t t0 = b * a.control_block;
  const int blk_len = (int)((a.n_samples - t0) < a.control_block ? (a.n_samples - t0) : a.control_block);
  float *xs = xs_all[wave];  // xs[32 + n] = detector input n of the block, xs[0..31] = the 32 before it
  const float *src = a.audio + (int64_t)s * a.stream_stride + t0;
  if (lane < kTpTaps) {
    float h = b == 0 ? a.st32[(int64_t)(kTpOutHist + lane) * a.n_streams + s] : src[lane - kTpTaps];
    xs[lane] = finite_f32(h) ? h : 0.0f;  // TruePeakDetector::process_block feeds 0 for a non-finite sample
  }
  float peak = 0.0f;
  const int padded = (blk_len + 63) & ~63;
  for (int n = lane; n < padded; n += 64) {
    float v = n < blk_len ? src[n] : 0.0f;
    v = finite_f32(v) ? v : 0.0f;
    xs[kTpTaps + n] = v;
    peak = fmaxf(peak, fabsf(v));
  }
  float bq[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) bq[k] = tb.b[k][lane];
  __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave: its LDS writes are in order, this keeps the compiler honest
  __builtin_amdgcn_wave_barrier();

  const int row = lane & 15, kq = lane >> 4;
  const int col_j = lane & 3;  // column (p, j) = lane & 15 with j in the low two bits
  // A[i][t] = x[n0 + 4 i + 3 - t]; in xs coordinates sample n sits at xs[32 + n]
  const float *ap = xs + kTpTaps + 4 * row + 3 - kq;
  for (int n0 = 0; n0 < padded; n0 += 128) {
    tp_v4f acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const bool second = n0 + 64 < padded;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float a0 = ap[n0 - 4 * k];
      const float a1 = second ? ap[n0 + 64 - 4 * k] : 0.0f;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bq[k], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bq[k], acc1, 0, 0, 0);
    }
    // acc[r]: row = (lane >> 4) * 4 + r, i.e. sample n0 + 4 row + j
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * ((lane >> 4) * 4 + r) + col_j;
      if (n < blk_len) peak = fmaxf(peak, fabsf(acc0[r]));
      if (n + 64 < blk_len) peak = fmaxf(peak, fabsf(acc1[r]));
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) peak = fmaxf(peak, __shfl_xor(peak, off, 64));
  if (lane == 0) a.stats[b * a.n_streams + s].output_true_peak = peak;
}

// The detector's 32-sample history for the next segment: a launch of its own behind tp_detect_kernel, whose first-block
// waves still read the old rows.
__global__ __launch_bounds__(256) void tp_history_kernel(TpDetectArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (row, stream), streams fastest
  if (idx >= (int64_t)kTpTaps * a.n_streams) return;
  const int r = (int)(idx / a.n_streams), s = (int)(idx - (int64_t)r * a.n_streams);
  const float v = a.audio[(int64_t)s * a.stream_stride + a.n_samples - kTpTaps + r];
  a.st32[(int64_t)(kTpOutHist + r) * a.n_streams + s] = finite_f32(v) ? v : 0.0f;
}

static TpDetectTable make_table() {
  TpDetectTable t{};
  for (int k = 0; k < 9; ++k)
    for (int lane = 0; lane < 64; ++lane) {
      const int kq = lane >> 4, col = lane & 15, p = col >> 2, j = col & 3;
      const int tap = j - 3 + 4 * k + kq;
      t.b[k][lane] = (tap >= 0 && tap < kTpTaps) ? AF_TP_FIR[p][tap] : 0.0f;
    }
  return t;
}

bool tp_detect_supported(int control_block, int64_t n_samples) {
  return control_block >= kTpTaps && control_block <= kTpMaxBlock && n_samples >= kTpTaps;
}

hipError_t launch_tp_detect(const float *audio, float *st32, BlockStats *stats, int64_t stream_stride, int64_t n_samples,
                            int32_t n_streams, int32_t control_block, hipStream_t stream) {
  static const TpDetectTable table = make_table();
  TpDetectArgs a{audio, st32, stats, stream_stride, n_samples, n_streams, control_block};
  const int64_t units = ((n_samples + control_block - 1) / control_block) * n_streams;
  if (units == 0) return hipSuccess;
  hipLaunchKernelGGL(tp_detect_kernel, dim3((unsigned)((units + kTpWavesPerGroup - 1) / kTpWavesPerGroup)),
                     dim3(64 * kTpWavesPerGroup), 0, stream, a, table);
  hipLaunchKernelGGL(tp_history_kernel, dim3((unsigned)(((int64_t)kTpTaps * n_streams + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace af
