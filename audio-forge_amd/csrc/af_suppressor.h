// af_suppressor.h -- data shared between the host side and the kernels of the RNNoise suppressor
// (rust-core/src/dsp/rnnoise.rs over nnnoiseless 0.5.2; see af_rnnoise.hip for the parity note).
#pragma once
#include <stdint.h>

namespace af {

constexpr int kRnnFrame = 480;
constexpr int kRnnWindow = 960;
constexpr int kRnnFreq = 481;
constexpr int kRnnBands = 22;
constexpr int kRnnFeat = 42;
constexpr int kRnnFeatPad = 44;
constexpr int kPitchMin = 60;
constexpr int kPitchMax = 768;
constexpr int kPitchBuf = 1728;
constexpr int kCepsMem = 8;

// ---- per-stream persistent state, one row of `SuppState::kCount` floats per stream -------------
struct SuppState {
  enum : int {
    kHist = 0,                          // last 1728 high-passed model-input samples (pitch_buf)
    kHpMem = kHist + kPitchBuf,         // 2
    kLastPeriod = kHpMem + 2,           // int stored as float
    kLastGain,
    kMemId,                             // cepstral ring write index (int as float)
    kCeps,                              // 8 x 22
    kLastG = kCeps + kCepsMem * kRnnBands,  // 22
    kVadState = kLastG + kRnnBands,     // 24
    kNoiseState = kVadState + 24,       // 48
    kDenoiseState = kNoiseState + 48,   // 96
    kSynthMem = kDenoiseState + 96,     // 480
    kSmoothedStrength = kSynthMem + kRnnFrame,
    kCount = ((kSmoothedStrength + 1 + 3) / 4) * 4
  };
};

// ---- per (frame, stream) workspace record -------------------------------------------------------
struct SuppFrameRec {
  float Ex[kRnnBands], Ep[kRnnBands], Exp[kRnnBands];
  float feat[kRnnFeatPad];
  float gains_raw[kRnnBands];  // network output (what pitch_filter sees)
  float gains[kRnnBands];      // after g = max(g, 0.6 lastg)
  int32_t silence;
  int32_t pitch_index;
};

// ---- network weights on the device: f32, padded to the 16x16x4 matrix-core tiles ------------------
// Each matrix is [K_pad][N_pad] row-major (k = input index, n = output unit); GRU layers hold three
// of them (z, r, h) over the concatenated input [layer input | recurrent state].
struct RnnLayerDims { int k_in, k_rec, k_pad, n, n_pad; };
constexpr RnnLayerDims kDimDense{42, 0, 44, 24, 32};
constexpr RnnLayerDims kDimVad{24, 24, 48, 24, 32};
constexpr RnnLayerDims kDimNoise{90, 48, 140, 48, 48};
constexpr RnnLayerDims kDimDenoise{114, 96, 212, 96, 96};
constexpr RnnLayerDims kDimOut{96, 0, 96, 22, 32};

// the eleven matrices in blob order: dense, vad z r h, noise z r h, denoise z r h, out
constexpr RnnLayerDims kRnnMatrixDims[11] = {kDimDense, kDimVad, kDimVad, kDimVad, kDimNoise, kDimNoise, kDimNoise,
                                             kDimDenoise, kDimDenoise, kDimDenoise, kDimOut};
// "w4" operand order of the wave-private network kernel: dword [group g][tile][lane] of a matrix holds
// W[16 g + 4 j + (lane >> 4)][16 tile + (lane & 15)] in byte j (k beyond k_pad: zero); offsets in dwords
constexpr int w4_matrix_dwords(int i) { return ((kRnnMatrixDims[i].k_pad / 4 + 3) / 4) * (kRnnMatrixDims[i].n_pad / 16) * 64; }
constexpr int w4_matrix_offset(int i) {
  int o = 0;
  for (int j = 0; j < i; ++j) o += w4_matrix_dwords(j);
  return o;
}

struct RnnDeviceWeights {
  const float *dense_w, *dense_b;           // [44][32], [32]
  const float *vad_w[3], *vad_b[3];         // [48][32]
  const float *noise_w[3], *noise_b[3];     // [140][48]
  const float *den_w[3], *den_b[3];         // [212][96]
  const float *out_w, *out_b;               // [96][32]
  const float *tansig;                      // [201]
  // the same eleven matrices as int8 (the model's native precision) in the operand order of the network kernel,
  // which streams them from L2 (w4_matrix_offset / w4_matrix_dwords above)
  const uint32_t *w4;
};

struct SuppTables {
  const float *half_window;   // [480]
  const float *dct;           // [22*22]
  const float2 *twiddle;      // [960] exp(-2 pi i k / 960)
  const float *frac;          // [404] position of a bin inside its band: (float)j / (float)band_size
  const int32_t *band_of_bin; // [484] band index of a bin (bins >= 400 belong to no band: 21)
};

struct SuppArgs {
  const float *in;            // chain input, [stream][stride]
  float *out;                 // suppressor output, same layout
  float *xh;                  // [stream][1728 + n_frames*480] scaled + high-passed model input
  const float *xh_prev;       // the previous window's buffer (history source for the pre-pass), or null: use `state`
  int64_t xh_prev_stride;     // 1728 + its frame count * 480
  float2 *X, *P;              // [frame][stream][481]
  float *ds;                  // [frame][stream][864] LPC-whitened, 2x-decimated pitch buffer (pitch part 1 -> part 2)
  SuppFrameRec *rec;          // [frame][stream]
  float *state;               // [stream][SuppState::kCount]
  int64_t stream_stride;      // of `out`
  int64_t in_stride;          // of `in` (the engine's frame-assembly buffer when samples were pending, else = stream_stride)
  int32_t n_streams;
  int32_t n_frames;           // frames in this window
  int64_t frame0;             // first frame of the window inside `in`
  float strength;             // wet/dry target (rnnoise.rs:70-79)
  float smoothing_coeff;      // rnnoise.rs:45-51
  int32_t raw_protocol;       // 1: rnnoise_benchmark.rs scaling (clamp*32768, no mix)
  // optional realtime front end folded into the prefilter pass (routing.rs:802-843)
  int32_t front_clamp, front_dc, front_hp;
  double hp_b0, hp_b1, hp_b2, hp_a1, hp_a2;
  long long *dbg;             // optional: per-section shader-clock totals of workgroup 0 (development aid)
  double *chain_st64;         // chain state planes (pre-filter memories live there)
  float *chain_st32;
  int32_t f64_pre_z1, f32_dc_x1;  // field indices inside those planes
};

}  // namespace af
