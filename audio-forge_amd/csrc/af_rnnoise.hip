// af_rnnoise.hip -- the RNNoise suppressor stage (rust-core/src/dsp/rnnoise.rs) on gfx950.
//
// PARITY NOTE.  The wrapper (soft clip, x32768 scaling, /32768, smoothed wet/dry mix,
// rnnoise.rs:45-164) follows the reference text.  The core is `nnnoiseless 0.5.2`
// (Cargo.lock:605-613, call site rnnoise.rs:142-143): a crate that is not vendored in the reference
// checkout and whose trained weights are embedded in it.  What is built here is the published RNNoise
// algorithm that crate ports, validated against this repository's CPU restatement
// (oracle/af_rnnoise.c) on seeded synthetic weights in the real layer layout.  Parity against the
// crate itself is UNPINNED.
//
// Seven kernels per window of frames:
//   supp_prefilter_kernel  lane per stream: model-input scaling + RNNoise's 2nd-order high-pass over
//                          samples (a recurrence), 64x64 tiles transposed through LDS.
//   supp_spectrum_kernel   wave per (frame, stream): 960-point windowed transform, 22 band energies.
//   supp_pitch_kernel      wave per stream, frames in order: LPC-whitened 2x-decimated buffer, coarse +
//                          fine cross-correlation, octave-error removal (looks at the previous frame),
//                          cepstral history and the features that come from it.
//   supp_pitchspec_kernel  wave per (frame, stream): pitch-aligned transform, band correlation, its
//                          cepstral features.
//   supp_rnn_kernel        16 streams per workgroup, frames in order: dense(42->24), GRU24, GRU48,
//                          GRU96, dense(96->22) on the f32 matrix cores (v_mfma_f32_16x16x4_f32).
//                          north_star asks for bf16 MFMA; bf16 activations (8 significant bits)
//                          would put ~1e-3 relative error on every gain, two orders outside the 1e-5
//                          budget.  The f32-input MFMA is an exact k-ordered fmaf chain, which lets
//                          this stage match the CPU restatement, and at ~0.2 GFLOP per stream-second
//                          the network is three orders of magnitude below even that unit's rate.
//   supp_resynth_kernel    wave per (frame, stream): pitch comb filter, band-gain interpolation, inverse
//                          transform, synthesis window.
//   supp_overlap_kernel    wave per stream, frames in order: overlap-add, /32768, smoothed wet/dry mix.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "af_fft_consts.h"
#include "af_suppressor.h"

namespace af {

__constant__ int c_eband[kRnnBands] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40, 48, 60, 78, 100};

// ============================================================================== prefilter
__device__ __forceinline__ bool finite32(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }

// RNNoiseProcessor::scale_sample_for_model, rnnoise.rs:89-111
__device__ __forceinline__ float scale_for_model(float sample) {
  const float kScale = 32768.0f, kLimit = 32760.0f, kLimitUnit = 32760.0f / 32768.0f, kThr = 0.98f;
  const float kKnee = 1.0f - 0.98f;
  float v;
  if (!finite32(sample)) {
    v = 0.0f;
  } else {
    const float magnitude = fabsf(sample);
    if (magnitude <= kThr) {
      v = sample;
    } else {
      const float over = magnitude - kThr;
      const float compressed = over / (over + kKnee);
      const float softened = kThr + (kLimitUnit - kThr) * compressed;
      v = copysignf(fminf(softened, kLimitUnit), sample);
    }
  }
  const float scaled = v * kScale;
  return scaled < -kLimit ? -kLimit : (scaled > kLimit ? kLimit : scaled);
}

// test tap: the transfer function above, element-wise (the reference pins it in rnnoise.rs:335-352)
__global__ void supp_scale_probe_kernel(const float *in, float *out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = scale_for_model(in[i]);
}
hipError_t launch_scale_probe(const float *in, float *out, int64_t n, hipStream_t stream) {
  hipLaunchKernelGGL(supp_scale_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, out, n);
  return hipGetLastError();
}

// The pass is a chain of short per-sample recurrences (DC block and 80 Hz high-pass in f64; the model's own high-pass)
// around a feed-forward soft clip, so its duration is samples x (instructions on the longest recurrence) x ~8 cycles (one
// wave issues a dependent vector instruction every ~8 cycles: tools/probe/valu_latency.hip) whatever the lane count.  As one
// wave doing everything that was ~800 cycles per sample (2.4-3.0 ms per 8 880-sample window at ANY batch: what a small batch
// waited for).  Now a workgroup of six waves takes 64 streams and works as a pipeline over 64-sample tiles, one barrier per
// tile: wave 0 loads tile i (64 row loads, transposed into LDS), wave 1 runs the front end on tile i-1 (lane = stream), wave 2
// the soft clip on tile i-2 (lane = TIME: 64 independent samples per instruction), wave 3 the model's high-pass on tile i-3
// (lane = stream), waves 4 and 5 write the dry signal of tile i-2 and the model input of tile i-4 back as rows.
constexpr int kPreGroup = 64;
constexpr int kPreWaves = 6;
constexpr int kPreTile = 64 * (kPreGroup + 1);  // floats per LDS tile
template <bool kClamp, bool kDcHp, bool kRaw>
__global__ __launch_bounds__(64 * kPreWaves) void supp_prefilter_kernel(SuppArgs a) {
  extern __shared__ float pre_lds[];  // in[2], dry[2], sc[2], xh[2]
  constexpr bool kFront = kClamp || kDcHp;
  float *t_in = pre_lds, *t_dry = pre_lds + 2 * kPreTile, *t_sc = pre_lds + 4 * kPreTile, *t_xh = pre_lds + 6 * kPreTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s0 = blockIdx.x * kPreGroup;
  const int s = s0 + lane;
  const bool valid = s < a.n_streams;
  const int sc = valid ? s : a.n_streams - 1;
  const int64_t NS = a.n_streams;
  const int64_t n = (int64_t)a.n_frames * kRnnFrame;
  const int64_t xh_stride = kPitchBuf + n;
  float *st = a.state + (int64_t)sc * SuppState::kCount;
  const int64_t ntiles = (n + 63) / 64;
  auto at = [](float *tile, int t, int col) -> float & { return tile[t * (kPreGroup + 1) + col]; };
  auto tile_len = [&](int64_t ti) { return (int)((n - ti * 64) < 64 ? (n - ti * 64) : 64); };

  // history: the previous 1728 model-input samples go in front of the window -- the tail of the previous
  // window's buffer when there is one (its kernels may still be running), else what the last call saved
  for (int r = wave; r < kPreGroup; r += kPreWaves) {
    const int sr = (s0 + r) < a.n_streams ? (s0 + r) : a.n_streams - 1;
    const float *hist = a.xh_prev ? a.xh_prev + (int64_t)sr * a.xh_prev_stride + (a.xh_prev_stride - kPitchBuf)
                                  : a.state + (int64_t)sr * SuppState::kCount + SuppState::kHist;
#pragma unroll 9
    for (int i = lane; i < kPitchBuf; i += 64) a.xh[(int64_t)sr * xh_stride + i] = hist[i];
  }

  // per-role state
  float m0 = 0.0f, m1 = 0.0f, dc_x1 = 0.0f, dc_y1 = 0.0f;
  double z1 = 0.0, z2 = 0.0;
  if (wave == 3) {
    m0 = st[SuppState::kHpMem];
    m1 = st[SuppState::kHpMem + 1];
  }
  if (wave == 1 && kDcHp) {  // realtime front end state (chain planes): DC block x1/y1 (f32), 80 Hz high-pass z1/z2 (f64)
    dc_x1 = a.chain_st32[(int64_t)a.f32_dc_x1 * NS + sc];
    dc_y1 = a.chain_st32[(int64_t)(a.f32_dc_x1 + 1) * NS + sc];
    z1 = a.chain_st64[(int64_t)a.f64_pre_z1 * NS + sc];
    z2 = a.chain_st64[(int64_t)(a.f64_pre_z1 + 1) * NS + sc];
  }
  const float b0 = -2.0f, b1 = 1.0f, a0 = -1.99599f, a1 = 0.99600f;  // RNNoise input high-pass
  const double hb0 = a.hp_b0, hb1 = a.hp_b1, hb2 = a.hp_b2, ha1 = a.hp_a1, ha2 = a.hp_a2;
  const bool hp_on = a.front_hp != 0;

  for (int64_t it = 0; it < ntiles + 4; ++it) {
    if (wave == 0) {  // ---- load tile `it` (rows are clamped, not skipped: the 64 loads stay in flight together)
      const int64_t ti = it;
      if (ti < ntiles) {
        const int len = tile_len(ti);
        const int64_t col = a.frame0 * kRnnFrame + ti * 64 + (lane < len ? lane : len - 1);
        float v[kPreGroup];
#pragma unroll
        for (int r = 0; r < kPreGroup; ++r) {
          const int sr = (s0 + r) < a.n_streams ? (s0 + r) : a.n_streams - 1;
          v[r] = a.in[(int64_t)sr * a.in_stride + col];
        }
        float *tile = t_in + (ti & 1) * kPreTile;
#pragma unroll
        for (int r = 0; r < kPreGroup; ++r) at(tile, lane, r) = v[r];
      }
    } else if (wave == 1) {  // ---- front end on tile it - 1 (lane = stream): routing.rs:802-843
      const int64_t ti = it - 1;
      if (ti >= 0 && ti < ntiles) {
        const int len = tile_len(ti);
        float *src = t_in + (ti & 1) * kPreTile, *dst = t_dry + (ti & 1) * kPreTile;
#pragma unroll 8
        for (int t = 0; t < 64; ++t) {
          if (t < len) {
            float v = at(src, t, lane);
            if (kFront) {
              if (!finite32(v)) v = 0.0f;                                        // routing.rs:808-811
              if (kClamp) v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);        // routing.rs:822
              if (kDcHp) {                                                       // routing.rs:832-840
                const float o = v - dc_x1 + 0.995f * dc_y1;
                dc_x1 = v;
                dc_y1 = o;
                v = o;
                if (hp_on) {
                  const double xin = (double)o;
                  const double y = hb0 * xin + z1;
                  z1 = hb1 * xin - ha1 * y + z2;
                  z2 = hb2 * xin - ha2 * y;
                  v = (float)y;
                }
              }
            }
            at(dst, t, lane) = v;
          }
        }
      }
    } else if (wave == 2) {  // ---- model-input scaling of tile it - 2 (lane = time)
      const int64_t ti = it - 2;
      if (ti >= 0 && ti < ntiles) {
        float *src = t_dry + (ti & 1) * kPreTile, *dst = t_sc + (ti & 1) * kPreTile;
#pragma unroll 8
        for (int r = 0; r < kPreGroup; ++r) {
          float v = at(src, lane, r);
          if (kRaw) {  // bin/rnnoise_benchmark.rs:75-79
            v = (v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v)) * 32768.0f;
          } else {
            v = scale_for_model(v);
          }
          at(dst, lane, r) = v;
        }
      }
    } else if (wave == 3) {  // ---- the model's own high-pass on tile it - 3 (lane = stream)
      const int64_t ti = it - 3;
      if (ti >= 0 && ti < ntiles) {
        const int len = tile_len(ti);
        float *src = t_sc + (ti & 1) * kPreTile, *dst = t_xh + (ti & 1) * kPreTile;
#pragma unroll 8
        for (int t = 0; t < 64; ++t) {
          if (t < len) {
            const float v = at(src, t, lane);
            const float y = v + m0;
            m0 = m1 + (b0 * v - a0 * y);
            m1 = (b1 * v - a1 * y);
            at(dst, t, lane) = y;
          }
        }
      }
    } else if (wave == 4) {  // ---- the dry signal the wet/dry mix sees is the suppressor's input, i.e. the front end's output
      const int64_t ti = it - 2;
      if (kFront && ti >= 0 && ti < ntiles) {
        const int len = tile_len(ti);
        float *src = t_dry + (ti & 1) * kPreTile;
#pragma unroll 8
        for (int r = 0; r < kPreGroup; ++r) {
          const int sr = s0 + r;
          if (sr < a.n_streams && lane < len)
            a.out[(int64_t)sr * a.stream_stride + a.frame0 * kRnnFrame + ti * 64 + lane] = at(src, lane, r);
        }
      }
    } else {  // ---- the model input of tile it - 4
      const int64_t ti = it - 4;
      if (ti >= 0 && ti < ntiles) {
        const int len = tile_len(ti);
        float *src = t_xh + (ti & 1) * kPreTile;
#pragma unroll 8
        for (int r = 0; r < kPreGroup; ++r) {
          const int sr = s0 + r;
          if (sr < a.n_streams && lane < len) a.xh[(int64_t)sr * xh_stride + kPitchBuf + ti * 64 + lane] = at(src, lane, r);
        }
      }
    }
    __syncthreads();
  }
  if (valid && wave == 3) {
    st[SuppState::kHpMem] = m0;
    st[SuppState::kHpMem + 1] = m1;
  }
  if (valid && wave == 1 && kDcHp) {
    a.chain_st32[(int64_t)a.f32_dc_x1 * NS + s] = dc_x1;
    a.chain_st32[(int64_t)(a.f32_dc_x1 + 1) * NS + s] = dc_y1;
    a.chain_st64[(int64_t)a.f64_pre_z1 * NS + s] = z1;
    a.chain_st64[(int64_t)(a.f64_pre_z1 + 1) * NS + s] = z2;
  }
}

// ============================================================================== FFT-960 on one wave
// 960 = 15 x 64.  Lane n2 transforms x[64 n1 + n2] over n1 in registers as a 3 x 5 prime-factor DFT (no
// twiddles between the two), applies W_960^(n2 k1), and the fifteen 64-point transforms across lanes run as
// two passes of radix-8 butterflies through LDS (120 butterflies per pass, two per lane).
// out[k] = sum_n in[n] exp(-2 pi i k n / 960) * scale.
//
// The three transform kernels (spectrum, pitch spectrum, resynthesis) run as workgroups of FOUR waves, each wave on
// its own (stream, frames) unit with a private 8.6 KB transform buffer.  What the waves share, read-only after one
// workgroup barrier, are the tables: the per-lane twiddles W_960^(lane k1) and W_64^(q r), the analysis window, the
// band tables.  Kept per wave in registers (the first form of these kernels) the twiddles and the window cost 61
// VGPRs, the kernels needed 180-232 and ran two waves per SIMD, latency bound (waves parked 50-74 % of their cycles);
// with the tables in LDS they fit three waves per SIMD (measured on the spectrum kernel alone: 1.44 -> 0.94 ms per
// 50-frame window).  A wave only ever synchronises with itself: LDS instructions of one wave execute in issue order,
// so a wave-level fence (wait for its own LDS traffic, no compiler motion across) is all the "barrier" a private
// buffer needs.
__constant__ uint16_t c_blk_first[54] = {0, 4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128, 136, 144, 152, 160, 168, 176, 184, 192, 200, 208, 216, 224, 232, 240, 248, 256, 264, 272, 280, 288, 296, 304, 312, 320, 328, 336, 344, 352, 360, 368, 376, 384, 392};
__constant__ uint8_t c_blk_count[54] = {4, 4, 4, 4, 4, 4, 4, 4, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8};
__constant__ uint8_t c_seg_blk0[21] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 18, 21, 24, 28, 34, 43};
__constant__ uint8_t c_seg_nblk[21] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 3, 3, 4, 6, 9, 11};
constexpr int kBandBlocks = 54;
constexpr int kFftBuf = 15 * 72;  // fifteen 8 x 8 tiles with 9-element rows: conflict-free in both passes
constexpr int kFftWaves = 4;      // waves (independent units) per transform workgroup
#ifndef AF_FFT_PREFETCH
#define AF_FFT_PREFETCH 1
#endif
constexpr int kBandSkewLen = 404 + (404 >> 3) + 1;  // a per-bin array laid out with one pad word every eight bins

__device__ __forceinline__ int band_skew(int bin) { return bin + (bin >> 3); }

struct FftShared {
  float2 tw960[15 * 64];  // [k1][lane] = W_960^(lane k1)
  float2 tw64[8 * 8];     // [r][q]     = W_64^(q r)
  float win[kRnnFrame];   // half of the symmetric analysis / synthesis window
  float frac[kBandSkewLen];  // position of a bin inside its band, at band_skew(bin)
  int32_t band_of[484];
  float dct6[kRnnBands * 6];  // the first six columns of the DCT matrix, [row][column]
  uint16_t blk_first[64];
  uint8_t blk_count[64], seg_blk0[32], seg_nblk[32];
};

__device__ __forceinline__ float2 cmul(float2 x, float2 w) { return make_float2(x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x); }
__device__ __forceinline__ float2 cadd(float2 x, float2 y) { return make_float2(x.x + y.x, x.y + y.y); }
__device__ __forceinline__ float2 csub(float2 x, float2 y) { return make_float2(x.x - y.x, x.y - y.y); }
__device__ __forceinline__ float2 mulmj(float2 z) { return make_float2(z.y, -z.x); }  // z * (-i)

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in issue order; this keeps the compiler from moving them across and waits
  // for the writes to land before other lanes of the same wave read them
  __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void dft3(float2 x0, float2 x1, float2 x2, float2 &y0, float2 &y1, float2 &y2) {
  const float c = 0.86602540378443864676f;  // sin(pi/3)
  const float2 t = cadd(x1, x2), d = csub(x1, x2);
  y0 = cadd(x0, t);
  const float2 m = make_float2(x0.x - 0.5f * t.x, x0.y - 0.5f * t.y);
  const float2 sj = make_float2(d.y * c, -d.x * c);
  y1 = cadd(m, sj);
  y2 = csub(m, sj);
}
__device__ __forceinline__ void dft5(const float2 *x, float2 *y) {
  const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
  const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
  const float2 t1 = cadd(x[1], x[4]), t2 = cadd(x[2], x[3]), d1 = csub(x[1], x[4]), d2 = csub(x[2], x[3]);
  y[0] = make_float2(x[0].x + t1.x + t2.x, x[0].y + t1.y + t2.y);
  const float2 m1 = make_float2(x[0].x + c1 * t1.x + c2 * t2.x, x[0].y + c1 * t1.y + c2 * t2.y);
  const float2 m2 = make_float2(x[0].x + c2 * t1.x + c1 * t2.x, x[0].y + c2 * t1.y + c1 * t2.y);
  const float2 n1 = mulmj(make_float2(s1 * d1.x + s2 * d2.x, s1 * d1.y + s2 * d2.y));
  const float2 n2 = mulmj(make_float2(s2 * d1.x - s1 * d2.x, s2 * d1.y - s1 * d2.y));
  y[1] = cadd(m1, n1);
  y[4] = csub(m1, n1);
  y[2] = cadd(m2, n2);
  y[3] = csub(m2, n2);
}
// 8-point DFT, radix-2 decimation in time
__device__ __forceinline__ void dft8(const float2 *v, float2 *V) {
  const float h = 0.70710678118654752440f;
  const float2 a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]), a2 = cadd(v[2], v[6]), a3 = mulmj(csub(v[2], v[6]));
  const float2 a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]), a6 = cadd(v[3], v[7]), a7 = mulmj(csub(v[3], v[7]));
  const float2 b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = csub(a1, a3);
  const float2 b4 = cadd(a4, a6), b6 = csub(a4, a6), b5 = cadd(a5, a7), b7 = csub(a5, a7);
  const float2 w1 = make_float2((b5.x + b5.y) * h, (b5.y - b5.x) * h);   // b5 * W8^1
  const float2 w2 = mulmj(b6);                                           // b6 * W8^2
  const float2 w3 = make_float2((b7.y - b7.x) * h, -(b7.x + b7.y) * h);  // b7 * W8^3
  V[0] = cadd(b0, b4);
  V[4] = csub(b0, b4);
  V[1] = cadd(b1, w1);
  V[5] = csub(b1, w1);
  V[2] = cadd(b2, w2);
  V[6] = csub(b2, w2);
  V[3] = cadd(b3, w3);
  V[7] = csub(b3, w3);
}

// the shared tables of a transform workgroup; the caller crosses one workgroup barrier afterwards
__device__ __forceinline__ void fft_shared_init(FftShared &S, const SuppTables &tb, int tid, int nthreads) {
  for (int i = tid; i < 15 * 64; i += nthreads) S.tw960[i] = tb.twiddle[(i & 63) * (i >> 6)];
  for (int i = tid; i < 64; i += nthreads) S.tw64[i] = tb.twiddle[15 * (i & 7) * (i >> 3)];
  for (int i = tid; i < kRnnFrame; i += nthreads) S.win[i] = tb.half_window[i];
  for (int i = tid; i < 404; i += nthreads) S.frac[band_skew(i)] = tb.frac[i];
  for (int i = tid; i < 484; i += nthreads) S.band_of[i] = tb.band_of_bin[i];
  for (int i = tid; i < kRnnBands * 6; i += nthreads) S.dct6[i] = tb.dct[(i / 6) * kRnnBands + (i % 6)];
  if (tid < 54) {
    S.blk_first[tid] = c_blk_first[tid];
    S.blk_count[tid] = c_blk_count[tid];
  }
  if (tid < 21) {
    S.seg_blk0[tid] = c_seg_blk0[tid];
    S.seg_nblk[tid] = c_seg_nblk[tid];
  }
}

// In place: every pass reads its operands into registers, then writes over the same buffer (which must hold kFftBuf
// elements and belongs to this wave alone).  The result lands in a[0 .. 960).
__device__ __forceinline__ void fft960_wave(float2 *a, const FftShared &S, int lane, float scale) {
  float2 x[15];
#pragma unroll
  for (int n1 = 0; n1 < 15; ++n1) x[n1] = a[64 * n1 + lane];
  wave_lds_fence();
  {
    // n1 = (5 na + 3 nb) mod 15, k1 = (10 ka + 6 kb) mod 15: a plain 3 x 5 two-dimensional DFT
    float2 t[3][5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
      dft3(x[(3 * nb) % 15], x[(5 + 3 * nb) % 15], x[(10 + 3 * nb) % 15], t[0][nb], t[1][nb], t[2][nb]);
#pragma unroll
    for (int ka = 0; ka < 3; ++ka) {
      float2 y[5];
      dft5(t[ka], y);
#pragma unroll
      for (int kb = 0; kb < 5; ++kb) {
        const int k1 = (10 * ka + 6 * kb) % 15;
        a[k1 * 72 + lane] = cmul(y[kb], S.tw960[k1 * 64 + lane]);  // rows 72 apart: pass A's 8 x 8 reads of two rows hit disjoint banks
      }
      __builtin_amdgcn_sched_barrier(0);  // one group's twiddles at a time (fetched all at once they cost 30 registers)
    }
  }
  wave_lds_fence();
  // 64 = 8 x 8, n2 = 8 p + q, k2 = r + 8 t.  Pass A: cell (k1, q): 8-point DFT over p, times W_64^(q r).
  // A cell stays inside its row k1, and the lane's second cell sits eight rows further down: the two halves run one
  // after the other (half the live registers of reading both first).
  {
    const int q = lane & 7;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int id = lane + 64 * half, k1 = id >> 3;
      if (id < 120) {
        float2 v[8], V[8];
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) v[pp] = a[k1 * 72 + 8 * pp + q];
        wave_lds_fence();
        dft8(v, V);
        a[k1 * 72 + q] = V[0];
#pragma unroll
        for (int r = 1; r < 8; ++r) a[k1 * 72 + r * 9 + q] = cmul(V[r], S.tw64[r * 8 + q]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  wave_lds_fence();
  // Pass B: cell (k1, r): 8-point DFT over q; output bin k1 + 15 (r + 8 t).
  {
    float2 v0[8], v1[8];
    const int id1 = lane + 64;
    const int k1a = lane >> 3, r = lane & 7, k1b = id1 >> 3;
#pragma unroll
    for (int q = 0; q < 8; ++q) v0[q] = a[k1a * 72 + r * 9 + q];
    if (id1 < 120) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v1[q] = a[k1b * 72 + r * 9 + q];
    }
    wave_lds_fence();
    float2 V[8];
    dft8(v0, V);
#pragma unroll
    for (int t = 0; t < 8; ++t) a[k1a + 15 * (r + 8 * t)] = make_float2(V[t].x * scale, V[t].y * scale);
    if (id1 < 120) {
      dft8(v1, V);
#pragma unroll
      for (int t = 0; t < 8; ++t) a[k1b + 15 * (r + 8 * t)] = make_float2(V[t].x * scale, V[t].y * scale);
    }
  }
  wave_lds_fence();
}

// ---- evaluation orders shared with the CPU restatement (oracle/af_rnnoise.c, "pitch tools") ----------
// dot64: 64 interleaved partial sums (mul then add), xor-butterfly combine; every lane returns the total.
// The xor butterfly 32, 16, 8, 4, 2, 1 of a wave-wide sum (every lane ends with the total), register to register: the same
// additions in the same order as `acc + __shfl_xor(acc, off)`, which the compiler turns into six ds_bpermute_b32 round trips
// through the LDS crossbar (~100 cycles each, and an s_waitcnt that also drains the kernel's other LDS traffic).  Stages 32 and
// 16 are gfx950's v_permlane32_swap / v_permlane16_swap (both halves / neighbouring rows exchanged; a + b == b + a); stage 8 is
// DPP row_ror:8 (lane i ^ 8 exactly); from there the partial sums repeat with period 8 across the wave, so row_ror:4 delivers
// the value lane i ^ 4 holds, and stages 2 and 1 are quad permutes.
template <int kBegin, int kEnd, typename F>
__device__ __forceinline__ void static_for(F &&f) {  // f(std::integral_constant<int, k>) for k in [kBegin, kEnd): indices stay constants
  if constexpr (kBegin < kEnd) {
    f(std::integral_constant<int, kBegin>{});
    static_for<kBegin + 1, kEnd>(f);
  }
}
template <int kCtrl>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), kCtrl, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_allsum_xor(float acc) {
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
    acc = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
    acc = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  acc = acc + dpp_move<0x128>(acc);  // row_ror:8
  acc = acc + dpp_move<0x124>(acc);  // row_ror:4
  acc = acc + dpp_move<0x4E>(acc);   // quad_perm:[2,3,0,1]
  acc = acc + dpp_move<0xB1>(acc);   // quad_perm:[1,0,3,2]
  return acc;
}
__device__ __forceinline__ float wave_dot64(const float *x, const float *y, int n, int lane) {
  float acc = 0.0f;
#pragma unroll 4
  for (int i = lane; i < n; i += 64) acc = acc + x[i] * y[i];
  return wave_allsum_xor(acc);
}
// The same sum for a length known at compile time (every call site's is): written out, the reads carry immediate offsets and only
// the last, partial round is masked -- the counted loop spent three instructions per round on its bookkeeping beside the one
// multiply-add (the pitch tracker forms 33 such sums of 240 products per frame, the search 16 of up to 864).
template <int kN>
__device__ __forceinline__ float wave_dot64_n(const float *x, const float *y, int lane) {
  float acc = 0.0f;
  const float *xl = x + lane, *yl = y + lane;
#pragma unroll
  for (int k = 0; k < (kN + 63) / 64; ++k) {
    if (64 * k + 63 < kN) acc = acc + xl[64 * k] * yl[64 * k];
    else if (lane < kN - 64 * k) acc = acc + xl[64 * k] * yl[64 * k];
  }
  return wave_allsum_xor(acc);
}
// same with a stride-2 view of y (the 4x-decimated buffer is every second sample of the 2x one)
__device__ __forceinline__ float wave_dot64_sq_stride2(const float *y, int n, int lane) {
  float acc = 0.0f;
  for (int i = lane; i < n; i += 64) acc = acc + y[2 * i] * y[2 * i];
  return wave_allsum_xor(acc);
}

// compute_band_energy / compute_band_corr: band b = R_b + F_b, the rising ramp over band b-1's bins and the
// falling ramp over band b's bins.  Evaluation order (shared with the CPU restatement): every band segment is
// cut into blocks of 8 bins (54 blocks in all, one lane each), a block is summed left to right, and a
// segment's block sums are added in block order -- a dependent chain of 8 + 11 additions instead of 88.
// The caller leaves the per-bin products x.re p.re + x.im p.im of bins 0..399 in `prod` at band_skew(bin): a lane
// that walks its block of eight consecutive bins then meets a different bank than its neighbours (stride 9), where
// the plain layout made every one of these reads, and those of `frac`, an 8-way bank conflict.
// `scratch` is 128 floats of this wave's LDS.  Lanes 0..21 return band `lane`.
__device__ __forceinline__ float band_sums_wave(const float *prod, float *scratch, const FftShared &S, int lane) {
  wave_lds_fence();  // the products are in place
  float rise = 0.0f, fall = 0.0f;
  if (lane < kBandBlocks) {
    const int e0 = S.blk_first[lane], count = S.blk_count[lane];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < count) {
        const int idx = band_skew(e0 + j);
        const float tmp = prod[idx];
        const float fr = S.frac[idx];
        fall += (1 - fr) * tmp;
        rise += fr * tmp;
      }
    }
  }
  scratch[lane] = rise;
  scratch[64 + lane] = fall;
  wave_lds_fence();
  float band = 0.0f;
  if (lane < kRnnBands) {
    float r = 0.0f, f = 0.0f;
    if (lane > 0) {
      const int k0 = S.seg_blk0[lane - 1], nk = S.seg_nblk[lane - 1];
      for (int k = 0; k < nk; ++k) r += scratch[k0 + k];
    }
    if (lane < kRnnBands - 1) {
      const int k0 = S.seg_blk0[lane], nk = S.seg_nblk[lane];
      for (int k = 0; k < nk; ++k) f += scratch[64 + k0 + k];
    }
    band = r + f;
    if (lane == 0 || lane == kRnnBands - 1) band *= 2;
  }
  wave_lds_fence();  // the sums are read: `scratch` and `prod` may be written again
  return band;
}

// interp_band_gain value at one bin
__device__ __forceinline__ float interp_gain(const float *bandE, const FftShared &S, int bin) {
  if (bin >= (100 << 2)) return 0.0f;
  const int b = S.band_of[bin];
  const float f = S.frac[band_skew(bin)];
  return (1 - f) * bandE[b] + f * bandE[b + 1];
}

// wave-private LDS of a transform unit: the transform buffer; once the spectrum sits in its first 481 slots the rest
// of it serves as the per-bin product array, the block-sum scratch and a few 22-entry band vectors
struct FftUnitLds {
  float2 fa[kFftBuf];
  __device__ float *prod() { return reinterpret_cast<float *>(fa + kRnnFreq + 1); }
  __device__ float *scratch() { return prod() + kBandSkewLen + 1; }
  __device__ float *small() { return scratch() + 128; }  // 6 x 32 floats
};
static_assert((kRnnFreq + 1) * 2 + kBandSkewLen + 1 + 128 + 6 * 32 <= kFftBuf * 2, "the tail of the transform buffer holds the band work arrays");

// ============================================================================== analysis, part 1
// One wave per (stream, group of frames): window, forward transform, band energies.  Fully parallel.

// Frames handled by one wave of the frame-parallel kernels: the per-wave setup is shared by kFramesPerWave frames.
constexpr int kFramesPerWave = 10;  // measured with the butterfly transform: 1 -> 357 ms per bench step, 5 -> 338, 10 -> 336

// a lane's 15 window coefficients sit at i = lane + 64 j of the symmetric 960-point window
__device__ __forceinline__ float window_at(const FftShared &S, int i) { return S.win[i < kRnnFrame ? i : kRnnWindow - 1 - i]; }

extern "C" __global__ __launch_bounds__(64 * kFftWaves, 3) void supp_spectrum_kernel(SuppArgs a, SuppTables tb) {
  __shared__ FftShared S;
  __shared__ FftUnitLds U[kFftWaves];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  fft_shared_init(S, tb, tid, 64 * kFftWaves);
  __syncthreads();  // the only workgroup barrier: from here on every wave works on its own unit
  const int groups = (a.n_frames + kFramesPerWave - 1) / kFramesPerWave;
  const int64_t unit = (int64_t)blockIdx.x * kFftWaves + wave;
  if (unit >= (int64_t)a.n_streams * groups) return;
  const int s = (int)(unit / groups), fg = (int)(unit % groups);
  const int64_t n = (int64_t)a.n_frames * kRnnFrame;
  FftUnitLds &L = U[wave];
  float nxt[15];  // the next frame's samples travel while this frame is transformed
  auto fetch = [&](int f) {
    const float *pb = a.xh + (int64_t)s * (kPitchBuf + n) + (int64_t)(f + 1) * kRnnFrame;
#pragma unroll
    for (int j = 0; j < 15; ++j) nxt[j] = pb[kPitchBuf - kRnnWindow + lane + 64 * j];
  };
  const int f_end = (fg + 1) * kFramesPerWave < a.n_frames ? (fg + 1) * kFramesPerWave : a.n_frames;
  fetch(fg * kFramesPerWave);
  for (int f = fg * kFramesPerWave; f < f_end; ++f) {
    const int64_t cell = (int64_t)f * a.n_streams + s;
#pragma unroll
    for (int j = 0; j < 15; ++j) L.fa[lane + 64 * j] = make_float2(nxt[j] * window_at(S, lane + 64 * j), 0.0f);
#if AF_FFT_PREFETCH
    if (f + 1 < f_end) fetch(f + 1);
#endif
    wave_lds_fence();
    fft960_wave(L.fa, S, lane, 1.0f / kRnnWindow);
#if !AF_FFT_PREFETCH
    if (f + 1 < f_end) fetch(f + 1);
#endif
    float2 *Xg = a.X + cell * kRnnFreq;
    float *prod = L.prod();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = lane + 64 * j;
      if (i < kRnnFreq) {
        const float2 x = L.fa[i];
        Xg[i] = x;
        if (i < 400) prod[band_skew(i)] = x.x * x.x + x.y * x.y;
      }
    }
    const float ex = band_sums_wave(prod, L.scratch(), S, lane);
    if (lane < kRnnBands) a.rec[cell].Ex[lane] = ex;
  }
}

// ============================================================================== analysis, part 2
// One wave per stream, frames in order: pitch (the octave-error removal looks at the previous frame) and the
// cepstral history.  Everything it touches is small, so sixteen of these waves share a CU.
struct PitchLds {  // (6.2 KB: 4096 one-wave workgroups fit the chip in ONE round -- 25 per CU by LDS, 21.3 needed on 192 CUs; with the
                   // search's arrays still in here (round 2: 9.9 KB, 16 per CU) the kernel took two)
  float ds[kPitchBuf / 2];
  float ylk[(kPitchMax >> 1) + 4];
  float ceps[kCepsMem][kRnnBands];
  float Ex[kRnnBands], Ly[kRnnBands];
  float feat[kRnnFeatPad];
  float dist[kCepsMem][kCepsMem];
};

// find_best_pitch (pitch.c) over precomputed numerators / energy deltas, in two phases that together reproduce the
// sequential scan exactly:
//   1. the window-energy recurrence Syy <- max(1, Syy + da[i]) is walked in order (two dependent operations per lag;
//      every lane computes the same values and leaves Syy[i] in `syy`);
//   2. the running best two candidates only change at lags that pass `num * best_den[1] > best_num[1] * Syy`, so 64
//      lags are tested at once against the current pair, the FIRST passing lag is applied the way the sequential scan
//      would, and only the lags after it are tested again: every lag meets exactly the state the sequential scan
//      would have shown it.  (The first form walked all lags one at a time on every lane: ~35 instructions per lag,
//      441 lags per frame, most of this kernel's instructions.)
template <int MP>
__device__ __forceinline__ void best_pitch_scan(const float *numa, const float *da, float *syy, float Syy, int lane, int &bp0,
                                                int &bp1) {
#pragma unroll 8
  for (int i = 0; i < MP; ++i) {
    syy[i] = Syy;  // same address, same value from every lane
    Syy = fmaxf(1.0f, Syy + da[i]);
  }
  __syncthreads();
  float bn0 = -1, bn1 = -1, bd0 = 0, bd1 = 0;
  bp0 = 0;
  bp1 = 1;
#pragma unroll
  for (int base = 0; base < MP; base += 64) {
    const int i = base + lane;
    const float num = i < MP ? numa[i] : -1.0f;
    const float sy = i < MP ? syy[i] : 1.0f;
    unsigned long long todo = ~0ull;  // lanes whose lag comes after the last applied one
    for (;;) {
      const bool pass = num >= 0.0f && num * bd1 > bn1 * sy;
      const unsigned long long m = __ballot(pass) & todo;
      if (m == 0) break;
      const int k = __ffsll((long long)m) - 1;
      const float nk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(num), k));
      const float sk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sy), k));
      if (nk * bd0 > bn0 * sk) {
        bn1 = bn0; bd1 = bd0; bp1 = bp0;
        bn0 = nk; bd0 = sk; bp0 = base + k;
      } else {
        bn1 = nk; bd1 = sk; bp1 = base + k;
      }
      todo = k == 63 ? 0ull : ~0ull << (k + 1);
    }
  }
}

// ---- pitch, part 1: everything that depends on the frame alone (wave per (frame, stream), fully parallel):
// 2x decimation + LPC-4 whitening, coarse and fine cross-correlation search.  Leaves the whitened buffer and
// the candidate period for part 2.
struct alignas(16) PitchSearchLds {
  float ds[kPitchBuf / 2];
  float xc[304];
  float numa[304], da[304];
  float syy[304];
  float d4[kPitchBuf / 4];  // every second sample of ds: the 4x-decimated buffer, contiguous (stride-2 reads of ds are 2-way bank conflicts)
};
extern "C" __global__ __launch_bounds__(64, 4) void supp_pitchsearch_kernel(SuppArgs a, SuppTables tb) {
  __shared__ PitchSearchLds L;
  const int lane = threadIdx.x;
  const int s = (int)(blockIdx.x / a.n_frames), f = (int)(blockIdx.x % a.n_frames);
  const int64_t n = (int64_t)a.n_frames * kRnnFrame;
  const float *xh = a.xh + (int64_t)s * (kPitchBuf + n);
  const int64_t cell = (int64_t)f * a.n_streams + s;
  {
    const float *pb = xh + (int64_t)(f + 1) * kRnnFrame;  // pitch_buf after shifting frame f in = pb[0 .. 1728)
    SuppFrameRec *rec = a.rec + cell;
    // ---------------- pitch_downsample (pitch.c): 2x decimation, LPC-4 whitening
    for (int i = lane; i < kPitchBuf / 2; i += 64)
      L.ds[i] = i == 0 ? .5f * (.5f * pb[1] + pb[0]) : .5f * (.5f * (pb[2 * i - 1] + pb[2 * i + 1]) + pb[2 * i]);
    __syncthreads();
    float n0, n1, n2, n3, n4;
    {
      float ac[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) ac[k] = wave_dot64(L.ds + k, L.ds, kPitchBuf / 2 - k, lane);
      ac[0] *= 1.0001f;
      for (int i = 1; i <= 4; ++i) ac[i] -= ac[i] * (.008f * i) * (.008f * i);
      float lpc[4] = {0, 0, 0, 0};
      float error = ac[0];
      if (ac[0] != 0) {
        for (int i = 0; i < 4; ++i) {
          float rr = 0;
          for (int j = 0; j < i; ++j) rr += lpc[j] * ac[i - j];
          rr += ac[i + 1];
          const float r = -rr / error;
          lpc[i] = r;
          for (int j = 0; j < (i + 1) >> 1; ++j) {
            const float t1 = lpc[j], t2 = lpc[i - 1 - j];
            lpc[j] = t1 + r * t2;
            lpc[i - 1 - j] = t2 + r * t1;
          }
          error = error - r * r * error;
          if (error < .001f * ac[0]) break;
        }
      }
      float tmp = 1.0f;
      for (int i = 0; i < 4; ++i) {
        tmp = .9f * tmp;
        lpc[i] = lpc[i] * tmp;
      }
      const float c1 = .8f;
      n0 = lpc[0] + .8f;
      n1 = lpc[1] + c1 * lpc[0];
      n2 = lpc[2] + c1 * lpc[1];
      n3 = lpc[3] + c1 * lpc[2];
      n4 = c1 * lpc[3];
    }
    {
      // celt_fir5 with zero initial memory: y[i] = x[i] + n0 x[i-1] + ... + n4 x[i-5], in that order
      float yv[14];
      int cnt = 0;
      for (int i = lane; i < kPitchBuf / 2; i += 64, ++cnt) {
        float sum = L.ds[i];
        sum += n0 * (i >= 1 ? L.ds[i - 1] : 0.0f);
        sum += n1 * (i >= 2 ? L.ds[i - 2] : 0.0f);
        sum += n2 * (i >= 3 ? L.ds[i - 3] : 0.0f);
        sum += n3 * (i >= 4 ? L.ds[i - 4] : 0.0f);
        sum += n4 * (i >= 5 ? L.ds[i - 5] : 0.0f);
        yv[cnt] = sum;
      }
      __syncthreads();
      cnt = 0;
      for (int i = lane; i < kPitchBuf / 2; i += 64, ++cnt) L.ds[i] = yv[cnt];
    }
    __syncthreads();
    // ---------------- pitch_search(x_lp = ds + 384, y = ds, len 960, max_pitch 588)
    const int max_pitch = kPitchMax - 3 * kPitchMin;  // 588
    const float *x_lp = L.ds + (kPitchMax >> 1);
    int best0, best1;
    {
      // coarse: 4x decimated, 147 lags x 240 products (lane per lag, left-to-right order)
      constexpr int len = kRnnWindow >> 2, mp = (kPitchMax - 3 * kPitchMin) >> 2;
      for (int i = lane; i < kPitchBuf / 4; i += 64) L.d4[i] = L.ds[2 * i];
      __syncthreads();
      {
        // xcorr[lag] = sum_j x[j] y[j + lag] on the matrix cores.  Write lag = 16 c + i and j = 4 s + k - 16 c: then
        //   D[i][c] += A[i][k] B[k][c],  A[i][k] = y[4 s + k + i],  B[k][c] = x[4 s + k - 16 c] (0 outside the frame),
        // summed over the steps s = 0..95, visits every j in ascending order for each (i, c), one fused multiply-add
        // per term (v_mfma_f32_16x16x4_f32 accumulates its four k in order): exactly inner_prod_fma of the CPU
        // restatement.  A zero B entry leaves the accumulator unchanged.  ~4 vector instructions per step instead of the
        // 16 (and 10 LDS reads) of the lane-per-lag loop this replaces.
        typedef float v4f_ps __attribute__((ext_vector_type(4)));
        v4f_ps acc = {0.0f, 0.0f, 0.0f, 0.0f};
        const int col = lane & 15, kq = lane >> 4;
        const float *ap = L.d4 + kq + col;                    // y[4 s + k + i]
        const float *x4 = L.d4 + (kPitchMax >> 2);            // x[j] = x_lp[2 j]
        int xi = kq - 16 * col;                               // 4 s + k - 16 c at s = 0
#pragma unroll 4
        for (int s = 0; s < 96; ++s) {
          const float av = ap[4 * s];
          const bool in = (unsigned)xi < (unsigned)len;
          const float bv = in ? x4[in ? xi : 0] : 0.0f;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
          xi += 4;
        }
        // acc[r]: row i = (lane >> 4) * 4 + r, column c = lane & 15
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lag = 16 * col + kq * 4 + r;
          if (lag < mp) {
            const float sum = acc[r];
            const float x16 = sum * 1e-12f;
            L.numa[lag] = sum > 0 ? x16 * x16 : -1.0f;
            const float ya = L.d4[lag + len], yb = L.d4[lag];
            L.da[lag] = ya * ya - yb * yb;
          }
        }
      }
      float Syy0;
      {
        float acc = 0.0f;
        for (int i = lane; i < len; i += 64) acc = acc + L.d4[i] * L.d4[i];
        Syy0 = 1.0f + wave_allsum_xor(acc);
      }
      __syncthreads();
      static_assert(mp == 147, "coarse lag count");
      best_pitch_scan<mp>(L.numa, L.da, L.syy, Syy0, lane, best0, best1);
    }
    __syncthreads();
    {
      // fine: 2x decimated, only within +-2 of the two coarse candidates (at most ten lags)
      constexpr int len = kRnnWindow >> 1, mp = (kPitchMax - 3 * kPitchMin) >> 1;
      for (int i = lane; i < mp; i += 64) {
        L.xc[i] = 0.0f;
        L.numa[i] = -1.0f;
        const float ya = L.ds[i + len], yb = L.ds[i];
        L.da[i] = ya * ya - yb * yb;
      }
      __syncthreads();
      for (int c = 0; c < 2; ++c) {
        const int centre = 2 * (c == 0 ? best0 : best1);
        for (int i = centre - 2; i <= centre + 2; ++i) {
          if (i < 0 || i >= mp) continue;
          if (c == 1) {
            const int d0 = i - 2 * best0;
            if (d0 <= 2 && d0 >= -2) continue;  // already done for the first candidate
          }
          const float v = fmaxf(-1.0f, wave_dot64(x_lp, L.ds + i, len, lane));
          if (lane == 0) {
            L.xc[i] = v;
            const float x16 = v * 1e-12f;
            L.numa[i] = v > 0 ? x16 * x16 : -1.0f;
          }
        }
      }
      const float Syy0 = 1.0f + wave_dot64(L.ds, L.ds, len, lane);
      __syncthreads();
      best_pitch_scan<mp>(L.numa, L.da, L.syy, Syy0, lane, best0, best1);
    }
    int pitch_index;
    {
      int offset = 0;
      if (best0 > 0 && best0 < (max_pitch >> 1) - 1) {
        const float pa = L.xc[best0 - 1], pbv = L.xc[best0], pc = L.xc[best0 + 1];
        if ((pc - pa) > .7f * (pbv - pa)) offset = 1;
        else if ((pa - pc) > .7f * (pbv - pc)) offset = -1;
      }
      pitch_index = kPitchMax - (2 * best0 - offset);
    }
    __syncthreads();
    __syncthreads();
    float *dsg = a.ds + cell * (kPitchBuf / 2);
    for (int i = lane; i < kPitchBuf / 2; i += 64) dsg[i] = L.ds[i];
    if (lane == 0) rec->pitch_index = pitch_index;
  }
}

// ---- pitch, part 1, round 3: FOUR frames of one stream per workgroup (a wave each).  Consecutive frames' 1728-sample windows
// overlap by 1248 samples, so the workgroup fetches the span once -- coalesced 16-byte loads, all in flight together --
// into LDS and every wave decimates its frame from there (the one-wave form read 6.9 KB per frame through 42 scalar loads per
// lane, a few at a time: 28 GB per bench step, and most of a wave's life spent waiting for them).  After the decimation the
// span's LDS becomes the waves' scan arrays (one workgroup barrier); from there on a wave only ever synchronises with itself.
// Arithmetic and evaluation orders are the one-wave kernel's: the whitened buffers and pitch indices agree bit for bit.
template <int MP>
__device__ __forceinline__ void best_pitch_scan_wave(const float *numa, const float *da, float *syy, float Syy, int lane, int &bp0, int &bp1) {
#pragma unroll 8
  for (int i = 0; i < MP; ++i) {
    syy[i] = Syy;  // same address, same value from every lane
    Syy = fmaxf(1.0f, Syy + da[i]);
  }
  wave_lds_fence();
  float bn0 = -1, bn1 = -1, bd0 = 0, bd1 = 0;
  bp0 = 0;
  bp1 = 1;
#pragma unroll
  for (int base = 0; base < MP; base += 64) {
    const int i = base + lane;
    const float num = i < MP ? numa[i] : -1.0f;
    const float sy = i < MP ? syy[i] : 1.0f;
    unsigned long long todo = ~0ull;
    for (;;) {
      const bool pass = num >= 0.0f && num * bd1 > bn1 * sy;
      const unsigned long long m = __ballot(pass) & todo;
      if (m == 0) break;
      const int k = __ffsll((long long)m) - 1;
      const float nk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(num), k));
      const float sk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sy), k));
      if (nk * bd0 > bn0 * sk) {
        bn1 = bn0; bd1 = bd0; bp1 = bp0;
        bn0 = nk; bd0 = sk; bp0 = base + k;
      } else {
        bn1 = nk; bd1 = sk; bp1 = base + k;
      }
      todo = k == 63 ? 0ull : ~0ull << (k + 1);
    }
  }
}

constexpr int kPsFrames = 4;                                   // frames (waves) per workgroup
constexpr int kPsX0 = 255;                                     // first data word of PsScan::c.x4z (17 x 15 guard words below it)
constexpr int kPsRaw = kPitchBuf + (kPsFrames - 1) * kRnnFrame;  // samples of the shared span: 3168
struct PsScan {  // a wave's scan arrays: the coarse stage's (d4, numa, da, syy over 147 lags), then the fine stage's (294 lags)
  union {
    // x4z: x[j] at kPsX0 + j + (j >> 4), zeros on both sides: the coarse correlation's B operand reads x[4 s + k - 16 c] for every
    // step s and column c, 255 words below and 152 above the data for the lags that do not exist -- as zeros in memory they cost
    // nothing, as a bounds test they were six vector instructions per matrix instruction
    struct { float d4[kPitchBuf / 4]; float numa[152], da[152], syy[152]; float x4z[672]; } c;
    struct { float xc[304], numa[304], da[304], syy[304]; } f;
  };
};
struct alignas(16) PitchSearch4Lds {
  float ds[kPsFrames][kPitchBuf / 2];
  union {
    float raw[kPsRaw];
    PsScan scan[kPsFrames];
  };
};
// Development aid (make EXTRA=-DAF_PS_PROFILE): shader-cycle stamps at the phase boundaries of the pitch search, printed by wave
// 0 of workgroup 0 and of one workgroup in the middle of the grid.
#ifdef AF_PS_PROFILE
#define AF_PS_DECL long long ps_t[9]; ps_t[8] = clock64()
#define AF_PS_STAMP(k) ps_t[k] = clock64()
#define AF_PS_PRINT                                                                                                        \
  if (lane == 0 && wave == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2))                                          \
  printf("pitch search wg %u: span %lld | decimate %lld | autocorr+lpc %lld | whiten+store %lld | coarse xcorr %lld | coarse scan %lld | " \
         "fine xcorr %lld | fine scan %lld | total %lld cycles\n", blockIdx.x, ps_t[0] - ps_t[8], ps_t[1] - ps_t[0], ps_t[2] - ps_t[1],     \
         ps_t[3] - ps_t[2], ps_t[4] - ps_t[3], ps_t[5] - ps_t[4], ps_t[6] - ps_t[5], ps_t[7] - ps_t[6], ps_t[7] - ps_t[8])
#else
#define AF_PS_DECL
#define AF_PS_STAMP(k)
#define AF_PS_PRINT
#endif
extern "C" __global__ __launch_bounds__(64 * kPsFrames, 4) void supp_pitchsearch4_kernel(SuppArgs a, SuppTables tb) {
  __shared__ PitchSearch4Lds L;
  AF_PS_DECL;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int groups = (a.n_frames + kPsFrames - 1) / kPsFrames;
  const int s = (int)(blockIdx.x / groups), f0 = (int)(blockIdx.x % groups) * kPsFrames;
  const int f = f0 + wave;
  const bool live = f < a.n_frames;
  const int64_t n = (int64_t)a.n_frames * kRnnFrame;
  const float *xh = a.xh + (int64_t)s * (kPitchBuf + n);
  {
    // the span: pitch_buf of frame f0 + k starts 480 k samples into it
    const int frames_here = (a.n_frames - f0) < kPsFrames ? (a.n_frames - f0) : kPsFrames;
    const int span = kPitchBuf + (frames_here - 1) * kRnnFrame;  // a multiple of 4
    const float4 *src = reinterpret_cast<const float4 *>(xh + (int64_t)(f0 + 1) * kRnnFrame);
    float4 *dst = reinterpret_cast<float4 *>(L.raw);
    for (int i = tid; i < span / 4; i += 64 * kPsFrames) dst[i] = src[i];
  }
  __syncthreads();
  AF_PS_STAMP(0);  // span in LDS
  float *ds = L.ds[wave];
  if (live) {
    const float *pb = L.raw + wave * kRnnFrame;  // pitch_buf after shifting frame f in = pb[0 .. 1728)
    // ---------------- pitch_downsample (pitch.c): 2x decimation
    // (rounds written out: 864 = 13 x 64 + 32, the special first sample only in round 0)
    static_for<0, (kPitchBuf / 2 + 63) / 64>([&](auto r_tag) {
      constexpr int r = decltype(r_tag)::value;
      const int i = lane + 64 * r;
      if (64 * r + 63 < kPitchBuf / 2 || lane < kPitchBuf / 2 - 64 * r) {
        if (r == 0) ds[i] = i == 0 ? .5f * (.5f * pb[1] + pb[0]) : .5f * (.5f * (pb[2 * i - 1] + pb[2 * i + 1]) + pb[2 * i]);
        else ds[i] = .5f * (.5f * (pb[2 * i - 1] + pb[2 * i + 1]) + pb[2 * i]);
      }
    });
  }
  __syncthreads();  // every wave has left the span: its LDS now holds the scan arrays
  if (!live) return;
  AF_PS_STAMP(1);  // decimated
  PsScan &S = L.scan[wave];
  const int64_t cell = (int64_t)f * a.n_streams + s;
  SuppFrameRec *rec = a.rec + cell;
  // ---------------- LPC-4 whitening
  float n0, n1, n2, n3, n4;
  {
    float ac[5];
    static_for<0, 5>([&](auto k_tag) {
      constexpr int k = decltype(k_tag)::value;
      ac[k] = wave_dot64_n<kPitchBuf / 2 - k>(ds + k, ds, lane);
    });
    ac[0] *= 1.0001f;
    for (int i = 1; i <= 4; ++i) ac[i] -= ac[i] * (.008f * i) * (.008f * i);
    float lpc[4] = {0, 0, 0, 0};
    float error = ac[0];
    if (ac[0] != 0) {
      for (int i = 0; i < 4; ++i) {
        float rr = 0;
        for (int j = 0; j < i; ++j) rr += lpc[j] * ac[i - j];
        rr += ac[i + 1];
        const float r = -rr / error;
        lpc[i] = r;
        for (int j = 0; j < (i + 1) >> 1; ++j) {
          const float t1 = lpc[j], t2 = lpc[i - 1 - j];
          lpc[j] = t1 + r * t2;
          lpc[i - 1 - j] = t2 + r * t1;
        }
        error = error - r * r * error;
        if (error < .001f * ac[0]) break;
      }
    }
    float tmp = 1.0f;
    for (int i = 0; i < 4; ++i) {
      tmp = .9f * tmp;
      lpc[i] = lpc[i] * tmp;
    }
    const float c1 = .8f;
    n0 = lpc[0] + .8f;
    n1 = lpc[1] + c1 * lpc[0];
    n2 = lpc[2] + c1 * lpc[1];
    n3 = lpc[3] + c1 * lpc[2];
    n4 = c1 * lpc[3];
  }
  AF_PS_STAMP(2);  // autocorrelation + LPC
  {
    // celt_fir5 with zero initial memory: y[i] = x[i] + n0 x[i-1] + ... + n4 x[i-5], in that order
    constexpr int kRounds = (kPitchBuf / 2 + 63) / 64;  // 14, the last one half a wave
    float yv[kRounds];
    static_for<0, kRounds>([&](auto r_tag) {
      constexpr int r = decltype(r_tag)::value;
      const int i = lane + 64 * r;
      float sum = 0.0f;
      if (64 * r + 63 < kPitchBuf / 2 || lane < kPitchBuf / 2 - 64 * r) {
        sum = ds[i];
        if (r == 0) {  // (the filter's zero initial memory only shows in the first five samples)
          sum += n0 * (i >= 1 ? ds[i >= 1 ? i - 1 : 0] : 0.0f);
          sum += n1 * (i >= 2 ? ds[i >= 2 ? i - 2 : 0] : 0.0f);
          sum += n2 * (i >= 3 ? ds[i >= 3 ? i - 3 : 0] : 0.0f);
          sum += n3 * (i >= 4 ? ds[i >= 4 ? i - 4 : 0] : 0.0f);
          sum += n4 * (i >= 5 ? ds[i >= 5 ? i - 5 : 0] : 0.0f);
        } else {
          sum += n0 * ds[i - 1];
          sum += n1 * ds[i - 2];
          sum += n2 * ds[i - 3];
          sum += n3 * ds[i - 4];
          sum += n4 * ds[i - 5];
        }
      }
      yv[r] = sum;
    });
    wave_lds_fence();
    // the whitened buffer goes back to LDS and out to memory (the pitch tracker reads it) in one pass, while the search runs
    float *dsg = a.ds + cell * (kPitchBuf / 2);
    static_for<0, kRounds>([&](auto r_tag) {
      constexpr int r = decltype(r_tag)::value;
      const int i = lane + 64 * r;
      if (64 * r + 63 < kPitchBuf / 2 || lane < kPitchBuf / 2 - 64 * r) {
        ds[i] = yv[r];
        dsg[i] = yv[r];
      }
    });
  }
  wave_lds_fence();
  AF_PS_STAMP(3);  // whitened + stored
  // ---------------- pitch_search(x_lp = ds + 384, y = ds, len 960, max_pitch 588)
  const int max_pitch = kPitchMax - 3 * kPitchMin;  // 588
  const float *x_lp = ds + (kPitchMax >> 1);
  int best0, best1;
  {
    // coarse: 4x decimated, 147 lags x 240 products on the matrix cores (see the one-wave kernel above)
    constexpr int len = kRnnWindow >> 2, mp = (kPitchMax - 3 * kPitchMin) >> 2;
    static_assert(kPsX0 + (len - 1) + ((len - 1) >> 4) < 672 && 3 + 4 * 95 + 23 + kPsX0 < 672, "x4z holds the data and both guards");
    for (int i = lane; i < kPsX0; i += 64) S.c.x4z[i] = 0.0f;
    for (int i = kPsX0 + len + ((len - 1) >> 4) + lane; i < 672; i += 64) S.c.x4z[i] = 0.0f;
    for (int i = lane; i < kPitchBuf / 4; i += 64) {
      const float v = ds[2 * i];
      S.c.d4[i] = v;
      // the B operand reads x[4 s + k - 16 c]: sixteen columns 16 apart -- two banks for a whole half-wave in the plain layout
      // (most of this kernel's 0.52 conflict / LDS-active ratio); one pad word per sixteen makes the column stride 17
      const int j = i - (kPitchMax >> 2);
      if (j >= 0 && j < len) S.c.x4z[kPsX0 + j + (j >> 4)] = v;
    }
    wave_lds_fence();
    {
      typedef float v4f_ps __attribute__((ext_vector_type(4)));
      v4f_ps acc = {0.0f, 0.0f, 0.0f, 0.0f};
      const int col = lane & 15, kq = lane >> 4;
      const float *ap = S.c.d4 + kq + col;          // y[4 s + k + i]
      // x[j] = x_lp[2 j] sits at kPsX0 + j + (j >> 4); j = 4 s + k - 16 c, k < 4, so j + floor(j / 16) = k - 17 c + 4 s + (s >> 2):
      // a per-lane base and a compile-time offset per step
      const float *bp = S.c.x4z + kPsX0 + kq - 17 * col;
#pragma unroll
      for (int st = 0; st < 96; ++st) {
        const float av = ap[4 * st];
        const float bv = bp[4 * st + (st >> 2)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lag = 16 * col + kq * 4 + r;
        if (lag < mp) {
          const float sum = acc[r];
          const float x16 = sum * 1e-12f;
          S.c.numa[lag] = sum > 0 ? x16 * x16 : -1.0f;
          const float ya = S.c.d4[lag + len], yb = S.c.d4[lag];
          S.c.da[lag] = ya * ya - yb * yb;
        }
      }
    }
    AF_PS_STAMP(4);  // coarse correlation
    float Syy0;
    {
      float acc = 0.0f;
      for (int i = lane; i < len; i += 64) acc = acc + S.c.d4[i] * S.c.d4[i];
      Syy0 = 1.0f + wave_allsum_xor(acc);
    }
    wave_lds_fence();
    static_assert(mp == 147, "coarse lag count");
    best_pitch_scan_wave<mp>(S.c.numa, S.c.da, S.c.syy, Syy0, lane, best0, best1);
  }
  AF_PS_STAMP(5);  // coarse scan
  wave_lds_fence();  // the coarse arrays are dead: the fine stage's take their place
  {
    // fine: 2x decimated, only within +-2 of the two coarse candidates (at most ten lags)
    constexpr int len = kRnnWindow >> 1, mp = (kPitchMax - 3 * kPitchMin) >> 1;
    for (int i = lane; i < mp; i += 64) {
      S.f.xc[i] = 0.0f;
      S.f.numa[i] = -1.0f;
      const float ya = ds[i + len], yb = ds[i];
      S.f.da[i] = ya * ya - yb * yb;
    }
    wave_lds_fence();
    for (int c = 0; c < 2; ++c) {
      const int centre = 2 * (c == 0 ? best0 : best1);
      for (int i = centre - 2; i <= centre + 2; ++i) {
        if (i < 0 || i >= mp) continue;
        if (c == 1) {
          const int d0 = i - 2 * best0;
          if (d0 <= 2 && d0 >= -2) continue;  // already done for the first candidate
        }
        const float v = fmaxf(-1.0f, wave_dot64_n<len>(x_lp, ds + i, lane));
        if (lane == 0) {
          S.f.xc[i] = v;
          const float x16 = v * 1e-12f;
          S.f.numa[i] = v > 0 ? x16 * x16 : -1.0f;
        }
      }
    }
    AF_PS_STAMP(6);  // fine correlations
    const float Syy0 = 1.0f + wave_dot64_n<len>(ds, ds, lane);
    wave_lds_fence();
    best_pitch_scan_wave<mp>(S.f.numa, S.f.da, S.f.syy, Syy0, lane, best0, best1);
  }
  int pitch_index;
  {
    int offset = 0;
    if (best0 > 0 && best0 < (max_pitch >> 1) - 1) {
      const float pa = S.f.xc[best0 - 1], pbv = S.f.xc[best0], pc = S.f.xc[best0 + 1];
      if ((pc - pa) > .7f * (pbv - pa)) offset = 1;
      else if ((pa - pc) > .7f * (pbv - pc)) offset = -1;
    }
    pitch_index = kPitchMax - (2 * best0 - offset);
  }
  if (lane == 0) rec->pitch_index = pitch_index;
  AF_PS_STAMP(7);  // fine scan
  AF_PS_PRINT;
}

// ---- pitch, part 2: what looks at the previous frame (wave per stream, frames in order): octave-error removal
// (remove_doubling compares with the last period and gain), the cepstral ring and the features built on it.
extern "C" __global__ __launch_bounds__(64, 6) void supp_pitch_kernel(SuppArgs a, SuppTables tb) {
  __shared__ PitchLds L;
  const int lane = threadIdx.x;
  const int s = blockIdx.x;
  const int64_t n = (int64_t)a.n_frames * kRnnFrame;
  const float *xh = a.xh + (int64_t)s * (kPitchBuf + n);
  float *st = a.state + (int64_t)s * SuppState::kCount;
  int last_period = (int)st[SuppState::kLastPeriod];
  float last_gain = st[SuppState::kLastGain];
  int memid = (int)st[SuppState::kMemId];
  for (int i = lane; i < kCepsMem * kRnnBands; i += 64) (&L.ceps[0][0])[i] = st[SuppState::kCeps + i];
  __syncthreads();

  // The frame's inputs (whitened buffer, band energies, the search's pitch index) are fetched one frame ahead, into registers: the
  // loop is one dependent chain per stream, and an unhidden trip to HBM per frame was a tenth of it.
  constexpr int kDsRegs = (kPitchBuf / 2 + 63) / 64;  // 14
  float ds_next[kDsRegs];
  float ex_next = 0.0f;
  int pitch_next = 0;
  auto prefetch = [&](int f) {
    const SuppFrameRec *rn = a.rec + ((int64_t)f * a.n_streams + s);
    const float *dsg = a.ds + ((int64_t)f * a.n_streams + s) * (kPitchBuf / 2);
#pragma unroll
    for (int k2 = 0; k2 < kDsRegs; ++k2) {
      const int i = lane + 64 * k2;
      ds_next[k2] = i < kPitchBuf / 2 ? dsg[i] : 0.0f;
    }
    ex_next = lane < kRnnBands ? rn->Ex[lane] : 0.0f;
    pitch_next = rn->pitch_index;
  };
  if (a.n_frames > 0) prefetch(0);
  for (int f = 0; f < a.n_frames; ++f) {
    SuppFrameRec *rec = a.rec + ((int64_t)f * a.n_streams + s);
    if (lane < kRnnBands) L.Ex[lane] = ex_next;
#pragma unroll
    for (int k2 = 0; k2 < kDsRegs; ++k2) {
      const int i = lane + 64 * k2;
      if (i < kPitchBuf / 2) L.ds[i] = ds_next[k2];
    }
    int pitch_index = __builtin_amdgcn_readfirstlane(pitch_next);  // (the same word on every lane: keep what follows from it scalar)
    const float ex_mine = ex_next;  // band `lane`'s energy (lanes < 22)
    if (f + 1 < a.n_frames) prefetch(f + 1);
    __syncthreads();
    // ---------------- remove_doubling(ds, 768, 60, 960, &pitch_index, last_period, last_gain)
    float gain;
    {
      const int minperiod0 = kPitchMin;
      const int maxperiod = kPitchMax / 2, minperiod = kPitchMin / 2, N = kRnnWindow / 2;
      int T0 = pitch_index / 2;
      const int prev_period = last_period / 2;
      const float *x = L.ds + maxperiod;
      if (T0 >= maxperiod) T0 = maxperiod - 1;
      const float xx = wave_dot64_n<N>(x, x, lane);
      float xy = wave_dot64_n<N>(x, x - T0, lane);
      {
        // yy_lookup[i] = max(0, xx + prefix_i), prefix over e_i = x[-i]^2 - x[N-i]^2 in the blocked scan order
        const int chunk = (maxperiod + 63) / 64;  // 6
        float local[6];
        float acc = 0.0f;
#pragma unroll
        for (int k2 = 0; k2 < 6; ++k2) {
          const int i = lane * chunk + k2 + 1;  // 1-based lag
          if (i <= maxperiod) acc = acc + (x[-i] * x[-i] - x[N - i] * x[N - i]);
          local[k2] = acc;
        }
        float total = acc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const float other = __shfl_up(total, off);
          if (lane >= off) total = total + other;
        }
        const float before = __shfl_up(total, 1);
        if (lane == 0) L.ylk[0] = xx;
#pragma unroll
        for (int k2 = 0; k2 < 6; ++k2) {
          const int i = lane * chunk + k2 + 1;
          if (i <= maxperiod) L.ylk[i] = fmaxf(0.0f, xx + (lane == 0 ? local[k2] : before + local[k2]));
        }
      }
      __syncthreads();
      float yy = L.ylk[T0];
      float best_xy = xy, best_yy = yy;
      const float g0 = xy / sqrtf(1 + xx * yy);
      float g = g0;
      int T = T0;
      // The candidates' lags depend on T0 alone, so all their correlations are formed first, as independent work (the decision
      // loop below is a dependent chain; with the two dot products inside it every one of them paid its LDS round trip alone).
      float xy_k[16], yy_k[16];
      int T1_k[16];
      static_for<2, 16>([&](auto k_tag) {
        constexpr int k2 = decltype(k_tag)::value;
        const int T1 = (2 * T0 + k2) / (2 * k2);
        int T1b;
        if (k2 == 2) T1b = (T1 + T0 > maxperiod) ? T0 : T0 + T1;
        else {
          constexpr int sc2 = (k2 == 6 || k2 == 12) ? 5 : ((k2 & 1) ? 2 : 3);  // second_check[k]
          T1b = (2 * sc2 * T0 + k2) / (2 * k2);
        }
        T1_k[k2] = T1;
        if (T1 >= minperiod) {  // (wave-uniform)
          // (every lane holds the same sums: parked in scalar registers, fourteen pairs of them would not fit the 80 vector ones)
          xy_k[k2] = __int_as_float(__builtin_amdgcn_readfirstlane(
              __float_as_int(.5f * (wave_dot64_n<N>(x, x - T1, lane) + wave_dot64_n<N>(x, x - T1b, lane)))));
          yy_k[k2] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(.5f * (L.ylk[T1] + L.ylk[T1b]))));
        } else {
          xy_k[k2] = 0.0f;
          yy_k[k2] = 0.0f;
        }
        if (k2 & 1) __builtin_amdgcn_sched_barrier(0);  // four dot products in flight at a time: more and the 80 registers spill
      });
      bool past_min = false;  // (the reference's loop breaks at the first candidate below the minimum period)
      static_for<2, 16>([&](auto k_tag) {
        constexpr int k2 = decltype(k_tag)::value;
        const int T1 = T1_k[k2];
        if (T1 < minperiod) past_min = true;
        if (past_min) return;
        xy = xy_k[k2];
        yy = yy_k[k2];
        const float g1 = xy / sqrtf(1 + xx * yy);
        float cont;
        const int dT = T1 - prev_period;
        if (dT <= 1 && dT >= -1) cont = last_gain;
        else if (dT <= 2 && dT >= -2 && 5 * k2 * k2 < T0) cont = .5f * last_gain;
        else cont = 0;
        float thresh = fmaxf(.3f, .7f * g0 - cont);
        if (T1 < 3 * minperiod) thresh = fmaxf(.4f, .85f * g0 - cont);
        else if (T1 < 2 * minperiod) thresh = fmaxf(.5f, .9f * g0 - cont);
        if (g1 > thresh) {
          best_xy = xy;
          best_yy = yy;
          T = T1;
          g = g1;
        }
      });
      best_xy = fmaxf(0.0f, best_xy);
      float pg = (best_yy <= best_xy) ? 1.0f : best_xy / (best_yy + 1);
      const float c0 = wave_dot64_n<N>(x, x - (T - 1), lane);
      const float c1v = wave_dot64_n<N>(x, x - T, lane);
      const float c2 = wave_dot64_n<N>(x, x - (T + 1), lane);
      int offset = 0;
      if ((c2 - c0) > .7f * (c1v - c0)) offset = 1;
      else if ((c0 - c2) > .7f * (c1v - c2)) offset = -1;
      if (pg > g) pg = g;
      pitch_index = 2 * T + offset;
      if (pitch_index < minperiod0) pitch_index = minperiod0;
      gain = pg;
    }
    last_period = pitch_index;
    last_gain = gain;
    // ---------------- features that do not need the pitch spectrum (denoise.c compute_frame_features)
    // column `lane` of the DCT matrix (lanes < 22), fetched here (L2 hits, hidden behind the logarithm and the scan) instead of
    // held across the frame: the candidate correlations above need the registers
    float dctcol[kRnnBands];
    {
      int col = lane < kRnnBands ? lane : 0;
      asm volatile("" : "+v"(col));  // (per frame: the compiler must not hoist the 22 loads out of the frame loop again)
#pragma unroll
      for (int j = 0; j < kRnnBands; ++j) dctcol[j] = tb.dct[j * kRnnBands + col];
    }
    float E = 0.0f;
    {
      // lane i takes band i's logarithm (every lane used to take all 22, ~40 instructions each: two fifths of this kernel's
      // instructions); the floor-following scan over the bands stays in band order, on values read lane by lane
      const float ly_raw = log10f(1e-2f + ex_mine);
      float logMax = -2, follow = -2, ly_mine = 0.0f;
#pragma unroll
      for (int i = 0; i < kRnnBands; ++i) {
        float ly = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ly_raw), i));
        ly = fmaxf(logMax - 7, fmaxf(follow - 1.5f, ly));
        if (lane == i) ly_mine = ly;
        logMax = fmaxf(logMax, ly);
        follow = fmaxf(follow - 1.5f, ly);
        E += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ex_mine), i));
      }
      if (lane < kRnnBands) L.Ly[lane] = ly_mine;
    }
    const bool silence = E < 0.04f;
    if (lane < kRnnFeatPad) L.feat[lane] = 0.0f;
    __syncthreads();
    if (!silence) {
      if (lane == 6) L.feat[kRnnBands + 18] = .01f * (pitch_index - 300);
      if (lane < kRnnBands) {  // dct(features, Ly)
        float sum = 0;
#pragma unroll
        for (int j = 0; j < kRnnBands; ++j) sum += L.Ly[j] * dctcol[j];
        float v = sum * sqrtf(2.0f / 22);
        if (lane == 0) v -= 12;
        if (lane == 1) v -= 4;
        L.feat[lane] = v;
        L.ceps[memid][lane] = v;
      }
      __syncthreads();
      const int m1 = memid < 1 ? kCepsMem + memid - 1 : memid - 1;
      const int m2 = memid < 2 ? kCepsMem + memid - 2 : memid - 2;
      if (lane < 6) {
        const float c0 = L.ceps[memid][lane], c1v = L.ceps[m1][lane], c2 = L.ceps[m2][lane];
        L.feat[lane] = c0 + c1v + c2;
        L.feat[kRnnBands + lane] = c0 - c2;
        L.feat[kRnnBands + 6 + lane] = c0 - 2 * c1v + c2;
      }
      memid = memid + 1 == kCepsMem ? 0 : memid + 1;
      {  // spectral variability: lane (i, j) owns one pair
        const int i = lane >> 3, j = lane & 7;
        float dist = 0;
        for (int k2 = 0; k2 < kRnnBands; ++k2) {
          const float t = L.ceps[i][k2] - L.ceps[j][k2];
          dist += t * t;
        }
        L.dist[i][j] = dist;
      }
      __syncthreads();
      {  // lane i < 8: the nearest other frame of the cepstral ring; the sum over i in order (one lane used to walk all 64 pairs)
        float mindist = 1e15f;
        if (lane < kCepsMem) {
#pragma unroll
          for (int j = 0; j < kCepsMem; ++j)
            if (j != lane) mindist = fminf(mindist, L.dist[lane][j]);
        }
        float spec_variability = 0;
#pragma unroll
        for (int i = 0; i < kCepsMem; ++i) spec_variability += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mindist), i));
        if (lane == 0) L.feat[kRnnBands + 19] = spec_variability / kCepsMem - 2.1f;
      }
    }
    __syncthreads();
    if (lane < kRnnFeatPad) rec->feat[lane] = L.feat[lane];
    if (lane == 0) {
      rec->silence = silence ? 1 : 0;
      rec->pitch_index = pitch_index;
    }
    __syncthreads();
  }
  if (lane == 0) {
    st[SuppState::kLastPeriod] = (float)last_period;
    st[SuppState::kLastGain] = last_gain;
    st[SuppState::kMemId] = (float)memid;
  }
  for (int i = lane; i < kCepsMem * kRnnBands; i += 64) st[SuppState::kCeps + i] = (&L.ceps[0][0])[i];
  // the last 1728 model-input samples become the next window's history
  for (int i = lane; i < kPitchBuf; i += 64) st[SuppState::kHist + i] = xh[n + i];
}

// ============================================================================== analysis, part 3
// One wave per (stream, group of frames): pitch-aligned transform, band energy / correlation, their cepstral features.
// The frame's spectrum X stays in registers (a lane owns bins lane + 64 j of X and of P alike).
extern "C" __global__ __launch_bounds__(64 * kFftWaves, 3) void supp_pitchspec_kernel(SuppArgs a, SuppTables tb) {
  __shared__ FftShared S;
  __shared__ FftUnitLds U[kFftWaves];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  fft_shared_init(S, tb, tid, 64 * kFftWaves);
  __syncthreads();
  const int groups = (a.n_frames + kFramesPerWave - 1) / kFramesPerWave;
  const int64_t unit = (int64_t)blockIdx.x * kFftWaves + wave;
  if (unit >= (int64_t)a.n_streams * groups) return;
  const int s = (int)(unit / groups), fg = (int)(unit % groups);
  const int64_t n = (int64_t)a.n_frames * kRnnFrame;
  FftUnitLds &L = U[wave];
  const int f_begin = fg * kFramesPerWave;
  const int f_end = (fg + 1) * kFramesPerWave < a.n_frames ? (fg + 1) * kFramesPerWave : a.n_frames;
  // the unit's pitch decisions come first: where the next frame's window starts depends on them
  int my_pitch = 0, my_silence = 0;
  if (f_begin + lane < f_end) {
    const SuppFrameRec *r = a.rec + ((int64_t)(f_begin + lane) * a.n_streams + s);
    my_pitch = r->pitch_index;
    my_silence = r->silence;
  }
  float nxt[15];
  auto fetch = [&](int f) {
    const int pitch_index = __builtin_amdgcn_readlane(my_pitch, f - f_begin);
    const float *pb = a.xh + (int64_t)s * (kPitchBuf + n) + (int64_t)(f + 1) * kRnnFrame;
#pragma unroll
    for (int j = 0; j < 15; ++j) nxt[j] = pb[kPitchBuf - kRnnWindow - pitch_index + lane + 64 * j];
  };
  fetch(f_begin);
  for (int f = f_begin; f < f_end; ++f) {
    const int64_t cell = (int64_t)f * a.n_streams + s;
    SuppFrameRec *rec = a.rec + cell;
    const bool silence = __builtin_amdgcn_readlane(my_silence, f - f_begin) != 0;
    const float2 *Xg = a.X + cell * kRnnFreq;
    float2 Xr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = lane + 64 * j;
      Xr[j] = i < kRnnFreq ? Xg[i] : make_float2(0.0f, 0.0f);
    }
    float ex = 0.0f;
    if (lane < kRnnBands) ex = rec->Ex[lane];
#pragma unroll
    for (int j = 0; j < 15; ++j) L.fa[lane + 64 * j] = make_float2(nxt[j] * window_at(S, lane + 64 * j), 0.0f);
#if AF_FFT_PREFETCH
    if (f + 1 < f_end) fetch(f + 1);
#endif
    wave_lds_fence();
    fft960_wave(L.fa, S, lane, 1.0f / kRnnWindow);
#if !AF_FFT_PREFETCH
    if (f + 1 < f_end) fetch(f + 1);
#endif
    float2 *Pg = a.P + cell * kRnnFreq;
    float *prod = L.prod();
    float2 Pr[7];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = lane + 64 * j;
      if (i < kRnnFreq) {
        const float2 p = L.fa[i];
        Pg[i] = p;
        if (j < 7) Pr[j] = p;
        if (i < 400) prod[band_skew(i)] = p.x * p.x + p.y * p.y;
      }
    }
    const float ep = band_sums_wave(prod, L.scratch(), S, lane);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int i = lane + 64 * j;
      if (i < 400) prod[band_skew(i)] = Xr[j].x * Pr[j].x + Xr[j].y * Pr[j].y;
    }
    float exp_ = band_sums_wave(prod, L.scratch(), S, lane);
    float *bandv = L.small();
    if (lane < kRnnBands) {
      exp_ = exp_ / sqrtf(.001f + ex * ep);
      rec->Ep[lane] = ep;
      rec->Exp[lane] = exp_;
      bandv[lane] = exp_;
    }
    wave_lds_fence();
    if (!silence && lane < 6) {  // dct(tmp, Exp), first six coefficients
      float sum = 0;
#pragma unroll
      for (int j = 0; j < kRnnBands; ++j) sum += bandv[j] * S.dct6[j * 6 + lane];
      float v = sum * sqrtf(2.0f / 22);
      if (lane == 0) v -= 1.3f;
      if (lane == 1) v -= 0.9f;
      rec->feat[kRnnBands + 12 + lane] = v;
    }
    wave_lds_fence();
  }
}

// ============================================================================== network
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float tansig_approx(const float *table, float x) {
  if (!(x < 8)) return 1;
  if (!(x > -8)) return -1;
  float sign = 1;
  if (x < 0) {
    x = -x;
    sign = -1;
  }
  const int i = (int)floorf(.5f + 25 * x);
  x -= .04f * i;
  float y = table[i];
  const float dy = 1 - y * y;
  y = y + x * dy * (1 - y * x);
  return sign * y;
}
__device__ __forceinline__ float sigmoid_approx(const float *table, float x) { return .5f + .5f * tansig_approx(table, .5f * x); }

// The first network kernel gave 16 streams to a 4-wave workgroup: every layer was a handful of tiles spread over
// the waves, with a barrier before and after (14 per frame) and copies to build each layer's concatenated input.
// A frame is a strictly dependent chain, so that kernel's duration was the sum of those latencies (matrix cores
// busy 5 %); and with 90 KB of weights + 56 KB of activations in LDS a workgroup needed a CU to itself, so beside
// the other suppressor kernels its workgroups mostly waited for a CU to drain.  Here ONE wave owns 16 streams for
// the whole window, needs 26 KB of LDS and never meets a workgroup barrier inside the frame loop:
//   * the recurrent states live in registers in the matrix-core result layout (row = stream, column = unit), so z,
//     the state, and the candidate h of a GRU unit meet in the same lane without touching LDS;
//   * the layer inputs are two LDS rows per stream laid out so that every concatenation the model needs is a
//     contiguous slice:  r1 = [dense | vad | features | noise]  (vad GRU reads r1[0:48], noise GRU all of it),
//     r2 = [vad | noise | features | denoise]  (denoise GRU; the output layer reads r2[114:210]);
//   * the int8 weights stream from L2 (90 KB in all, shared by every wave on the chip) as one dword per lane per
//     four k-steps (`w4` layout below), fetched three groups = 12 k-steps ahead of their use;
//   * several independent accumulator chains (tiles of one gate) run through one pass over K, sharing each A
//     operand, so the 44-cycle dependent latency of v_mfma_f32_16x16x4_f32 is hidden;
//   * the next frame's features are fetched while the current frame computes.
// Arithmetic per (stream, unit) is the same k-ordered fmaf chain from the bias as before: results are unchanged
// bit for bit (tools/ab_suppressor.py).
// Row strides of the two activation rows per stream.  A matrix step reads A[stream = lane & 15][k0 + (lane >> 4)]: within a
// half-wave sixteen streams x two k.  With the rows exactly as long as their contents (140 / 212 floats: 12 and 20 mod 32) streams
// c and c + 8 met in the same bank on EVERY read (LDS conflict cycles 1.05 x the LDS-active cycles in round 2's counters); a stride
// of 2 mod 32 spreads a half-wave's 32 reads over the 32 banks.  (The tails are never read: K_PAD ends inside the contents.)
constexpr int kR1Len = 140, kR2Len = 212;
constexpr int kR1 = 162, kR2 = 226;
static_assert(kR1 >= kR1Len && kR2 >= kR2Len && kR1 % 32 == 2 && kR2 % 32 == 2, "activation row strides");
constexpr int kR1Vad = 24, kR1Feat = 48, kR1Noise = 90;   // r1 = [dense 24 | vad 24 | features 42 | noise 48 | 0 0]
constexpr int kR2Noise = 24, kR2Feat = 72, kR2Den = 114;  // r2 = [vad 24 | noise 48 | features 42 | denoise 96 | 0 0]
constexpr int kRnnBiasTiles = 37;  // dense 2 | vad z r h 2 each | noise z r h 3 each | denoise z r h 6 each | out 2
constexpr int kRnnTablePad = 208;
constexpr int kW4Ahead = 3;        // weight groups in flight ahead of the one being consumed


// NC chains (tiles 0 .. NC - 1 of matrix MATRIX) through its K.  Weights come through a buffer descriptor over the
// whole w4 blob: address = descriptor base + lane * 4 (the one VGPR) + a compile-time scalar offset, so no load needs
// a pointer of its own (with flat addresses the compiler kept ~450 loop-invariant 64-bit pointers alive and spilled).
template <int NC, int MATRIX>
__device__ __forceinline__ void mfma_chains(const float *ap, __amdgpu_buffer_rsrc_t wsrc, unsigned lane4, v4f (&acc)[NC]) {
  constexpr int K_PAD = kRnnMatrixDims[MATRIX].k_pad, TILES = kRnnMatrixDims[MATRIX].n_pad / 16;
  static_assert(NC == TILES, "one chain per 16-unit tile");
  constexpr int kSteps = K_PAD / 4, kGroups = (kSteps + 3) / 4;
  constexpr int kBase = w4_matrix_offset(MATRIX);
  uint32_t wbuf[kW4Ahead][NC];
#pragma unroll
  for (int p = 0; p < kW4Ahead; ++p)
#pragma unroll
    for (int c = 0; c < NC; ++c)
      wbuf[p][c] = p < kGroups ? __builtin_amdgcn_raw_buffer_load_b32(wsrc, lane4, (kBase + (p * TILES + c) * 64) * 4, 0) : 0u;
  float av = ap[0];  // A[stream = lane & 15][k0 + (lane >> 4)], fetched one step ahead of its use
#pragma unroll
  for (int g = 0; g < kGroups; ++g) {
    uint32_t cur[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cur[c] = wbuf[g % kW4Ahead][c];
    if (g + kW4Ahead < kGroups) {
#pragma unroll
      for (int c = 0; c < NC; ++c)
        wbuf[g % kW4Ahead][c] =
            __builtin_amdgcn_raw_buffer_load_b32(wsrc, lane4, (kBase + ((g + kW4Ahead) * TILES + c) * 64) * 4, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int step = g * 4 + j;
      if (step < kSteps) {
        const float av_next = ap[(step + 1 < kSteps ? step + 1 : step) * 4];
        __builtin_amdgcn_sched_barrier(0);  // keep the fetches ahead of this step's matrix operations
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float wv = (float)(int)(int8_t)(cur[c] >> (8 * j));
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wv, acc[c], 0, 0, 0);
        }
        av = av_next;
      }
    }
  }
}

size_t rnn_lds_bytes(int waves) {
  return (kRnnTablePad + kRnnBiasTiles * 16) * sizeof(float) + (size_t)waves * 16 * (kR1 + kR2 + 32) * sizeof(float);
}

template <int NC>
__device__ __forceinline__ void load_bias(v4f (&acc)[NC], const float *bias, int first_bias_tile, int col) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float bv = bias[(first_bias_tile + c) * 16 + col];
    acc[c] = v4f{bv, bv, bv, bv};
  }
}

template <int kWaves>
__global__ __launch_bounds__(64 * kWaves, 2) void supp_rnn_kernel(SuppArgs a, RnnDeviceWeights w) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rnn_lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const unsigned lane = tid & 63;
  float *tansig = reinterpret_cast<float *>(rnn_lds);
  float *bias = tansig + kRnnTablePad;
  float *act = bias + kRnnBiasTiles * 16 + wave * (16 * (kR1 + kR2 + 32));
  float(*r1)[kR1] = reinterpret_cast<float(*)[kR1]>(act);
  float(*r2)[kR2] = reinterpret_cast<float(*)[kR2]>(act + 16 * kR1);
  float(*lastg)[32] = reinterpret_cast<float(*)[32]>(act + 16 * (kR1 + kR2));

  for (int i = tid; i < 201; i += 64 * kWaves) tansig[i] = w.tansig[i];
  for (int i = tid; i < kRnnBiasTiles * 16; i += 64 * kWaves) {
    const int t = i >> 4, c = i & 15;
    const float *src;
    int tt;
    if (t < 2) { src = w.dense_b; tt = t; }
    else if (t < 8) { src = w.vad_b[(t - 2) / 2]; tt = (t - 2) % 2; }
    else if (t < 17) { src = w.noise_b[(t - 8) / 3]; tt = (t - 8) % 3; }
    else if (t < 35) { src = w.den_b[(t - 17) / 6]; tt = (t - 17) % 6; }
    else { src = w.out_b; tt = t - 35; }
    bias[i] = src[tt * 16 + c];
  }
  __syncthreads();  // the only workgroup barrier: table and biases are in place

  const int s0 = (blockIdx.x * kWaves + wave) * 16;
  if (s0 >= a.n_streams) return;
  const int col = lane & 15, rq = lane >> 4;
  const float kScale = 1.f / 256;
  const int NS = a.n_streams;

  // ---- recurrent state in registers: element [r] of tile t belongs to stream rq*4 + r, unit t*16 + col
  float st_vad[2][4], st_noise[3][4], st_den[6][4];
  int srow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int s = s0 + rq * 4 + r;
    srow[r] = s < NS ? s : NS - 1;
    const float *st = a.state + (int64_t)srow[r] * SuppState::kCount;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int unit = t * 16 + col;
      st_vad[t][r] = unit < 24 ? st[SuppState::kVadState + unit] : 0.0f;
      lastg[rq * 4 + r][unit] = unit < kRnnBands ? st[SuppState::kLastG + unit] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) st_noise[t][r] = st[SuppState::kNoiseState + t * 16 + col];
#pragma unroll
    for (int t = 0; t < 6; ++t) st_den[t][r] = st[SuppState::kDenoiseState + t * 16 + col];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = rq * 4 + r;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int unit = t * 16 + col;
      if (unit < 24) {
        r1[row][unit] = 0.0f;
        r1[row][kR1Vad + unit] = st_vad[t][r];
        r2[row][unit] = st_vad[t][r];
      }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      r1[row][kR1Noise + t * 16 + col] = st_noise[t][r];
      r2[row][kR2Noise + t * 16 + col] = st_noise[t][r];
    }
#pragma unroll
    for (int t = 0; t < 6; ++t) r2[row][kR2Den + t * 16 + col] = st_den[t][r];
  }
  if (lane < 16) {
    r1[lane][138] = r1[lane][139] = 0.0f;
    r2[lane][210] = r2[lane][211] = 0.0f;
  }

  // ---- features of a frame: 16 streams x 42 values, element i = lane + 64 j
  constexpr int kFeatPerLane = (16 * kRnnFeat + 63) / 64;  // 11
  float pf[kFeatPerLane];
  int psil[4];
  auto fetch = [&](int f) {
#pragma unroll
    for (int j = 0; j < kFeatPerLane; ++j) {
      const int i = lane + 64 * j;
      const int row = i / kRnnFeat, c = i - row * kRnnFeat;
      pf[j] = 0.0f;
      if (row < 16) {
        const int s = s0 + row < NS ? s0 + row : NS - 1;
        pf[j] = a.rec[(int64_t)f * NS + s].feat[c];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) psil[r] = a.rec[(int64_t)f * NS + srow[r]].silence;
  };
  if (a.n_frames > 0) fetch(0);

  const float *a_dense = &r1[col][kR1Feat + rq];
  const float *a_r1 = &r1[col][rq];
  const float *a_r2 = &r2[col][rq];
  const float *a_out = &r2[col][kR2Den + rq];
  const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(w.w4), 0, w4_matrix_offset(11) * 4, 0x00020000);
  const unsigned lane4 = lane * 4;

  for (int f = 0; f < a.n_frames; ++f) {
    int sil[4];
#pragma unroll
    for (int j = 0; j < kFeatPerLane; ++j) {
      const int i = lane + 64 * j;
      const int row = i / kRnnFeat, c = i - row * kRnnFeat;
      if (row < 16) {
        r1[row][kR1Feat + c] = pf[j];
        r2[row][kR2Feat + c] = pf[j];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sil[r] = psil[r];
    wave_lds_fence();
    if (f + 1 < a.n_frames) fetch(f + 1);

    // ---- input_dense 42 -> 24 (tanh).  K runs over r1[48:92]: the two columns past the features hold noise-state
    // values, which meet the zero rows of the padded weight matrix
    {
      v4f acc[2];
      load_bias(acc, bias, 0, col);
      mfma_chains<2, 0>(a_dense, wsrc, lane4, acc);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int unit = c * 16 + col;
          if (unit < 24) r1[rq * 4 + r][unit] = tansig_approx(tansig, kScale * acc[c][r]);
        }
    }
    wave_lds_fence();

    // ---- vad GRU (24 in, 24 units) over r1[0:48]
    {
      float z[2][4];
      v4f acc[2];
      load_bias(acc, bias, 2, col);
      mfma_chains<2, 1>(a_r1, wsrc, lane4, acc);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) z[t][r] = sigmoid_approx(tansig, kScale * acc[t][r]);
      load_bias(acc, bias, 4, col);
      mfma_chains<2, 2>(a_r1, wsrc, lane4, acc);
      wave_lds_fence();  // every z / r chain has read the old state: its slots may hold the r-gated state now
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (t * 16 + col < 24) r1[rq * 4 + r][kR1Vad + t * 16 + col] = st_vad[t][r] * sigmoid_approx(tansig, kScale * acc[t][r]);
      wave_lds_fence();
      load_bias(acc, bias, 6, col);
      mfma_chains<2, 3>(a_r1, wsrc, lane4, acc);
      wave_lds_fence();
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sum = kScale * acc[t][r];
          sum = sum < 0 ? 0 : sum;
          const float h = z[t][r] * st_vad[t][r] + (1 - z[t][r]) * sum;
          if (!sil[r]) st_vad[t][r] = h;
          if (t * 16 + col < 24) {
            r1[rq * 4 + r][kR1Vad + t * 16 + col] = st_vad[t][r];
            r2[rq * 4 + r][t * 16 + col] = st_vad[t][r];
          }
        }
      wave_lds_fence();
    }

    // ---- noise GRU (90 in, 48 units) over all of r1
    {
      float z[3][4];
      v4f acc[3];
      load_bias(acc, bias, 8, col);
      mfma_chains<3, 4>(a_r1, wsrc, lane4, acc);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) z[t][r] = sigmoid_approx(tansig, kScale * acc[t][r]);
      load_bias(acc, bias, 11, col);
      mfma_chains<3, 5>(a_r1, wsrc, lane4, acc);
      wave_lds_fence();
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          r1[rq * 4 + r][kR1Noise + t * 16 + col] = st_noise[t][r] * sigmoid_approx(tansig, kScale * acc[t][r]);
      wave_lds_fence();
      load_bias(acc, bias, 14, col);
      mfma_chains<3, 6>(a_r1, wsrc, lane4, acc);
      wave_lds_fence();
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sum = kScale * acc[t][r];
          sum = sum < 0 ? 0 : sum;
          const float h = z[t][r] * st_noise[t][r] + (1 - z[t][r]) * sum;
          if (!sil[r]) st_noise[t][r] = h;
          r1[rq * 4 + r][kR1Noise + t * 16 + col] = st_noise[t][r];
          r2[rq * 4 + r][kR2Noise + t * 16 + col] = st_noise[t][r];
        }
      wave_lds_fence();
    }

    // ---- denoise GRU (114 in, 96 units) over all of r2: z (six chains), then r (six chains), then h
    {
      float z[6][4];
      v4f acc[6];
      load_bias(acc, bias, 17, col);
      mfma_chains<6, 7>(a_r2, wsrc, lane4, acc);
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) z[t][r] = sigmoid_approx(tansig, kScale * acc[t][r]);
      load_bias(acc, bias, 23, col);
      mfma_chains<6, 8>(a_r2, wsrc, lane4, acc);
      wave_lds_fence();
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          r2[rq * 4 + r][kR2Den + t * 16 + col] = st_den[t][r] * sigmoid_approx(tansig, kScale * acc[t][r]);
      wave_lds_fence();
      load_bias(acc, bias, 29, col);
      mfma_chains<6, 9>(a_r2, wsrc, lane4, acc);
      wave_lds_fence();
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sum = kScale * acc[t][r];
          sum = sum < 0 ? 0 : sum;
          const float h = z[t][r] * st_den[t][r] + (1 - z[t][r]) * sum;
          if (!sil[r]) st_den[t][r] = h;
          r2[rq * 4 + r][kR2Den + t * 16 + col] = st_den[t][r];
        }
      wave_lds_fence();
    }

    // ---- denoise_output 96 -> 22 (sigmoid) over r2[114:210], then g = max(g, 0.6 lastg)
    {
      v4f acc[2];
      load_bias(acc, bias, 35, col);
      mfma_chains<2, 10>(a_out, wsrc, lane4, acc);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int unit = t * 16 + col, s = s0 + rq * 4 + r;
          if (unit < kRnnBands && s < NS) {
            SuppFrameRec *rec = a.rec + ((int64_t)f * NS + s);
            if (!sil[r]) {
              float gv = sigmoid_approx(tansig, kScale * acc[t][r]);
              rec->gains_raw[unit] = gv;
              gv = fmaxf(gv, 0.6f * lastg[rq * 4 + r][unit]);
              lastg[rq * 4 + r][unit] = gv;  // only this lane ever touches the slot
              rec->gains[unit] = gv;
            } else {
              rec->gains[unit] = 1.0f;
              rec->gains_raw[unit] = 1.0f;
            }
          }
        }
    }
  }

  // ---- state back to HBM
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int s = s0 + rq * 4 + r;
    if (s >= NS) continue;
    float *st = a.state + (int64_t)s * SuppState::kCount;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int unit = t * 16 + col;
      if (unit < 24) st[SuppState::kVadState + unit] = st_vad[t][r];
      if (unit < kRnnBands) st[SuppState::kLastG + unit] = lastg[rq * 4 + r][unit];
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) st[SuppState::kNoiseState + t * 16 + col] = st_noise[t][r];
#pragma unroll
    for (int t = 0; t < 6; ++t) st[SuppState::kDenoiseState + t * 16 + col] = st_den[t][r];
  }
}

// ============================================================================== synthesis
// One wave per (stream, group of frames): comb filter, gains, inverse transform, synthesis window.  The 960 windowed
// samples of the frame overwrite the cell's P spectrum (no longer needed: 481 complex = 962 floats >= 960).
// X and P are elementwise work on a lane's own bins (lane + 64 j), so they live in registers; the transform buffer is
// free until the inverse transform and lends its space to the band sums.
extern "C" __global__ __launch_bounds__(64 * kFftWaves, 3) void supp_resynth_kernel(SuppArgs a, SuppTables tb) {
  __shared__ FftShared S;
  __shared__ FftUnitLds U[kFftWaves];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  fft_shared_init(S, tb, tid, 64 * kFftWaves);
  __syncthreads();
  const int groups = (a.n_frames + kFramesPerWave - 1) / kFramesPerWave;
  const int64_t unit = (int64_t)blockIdx.x * kFftWaves + wave;
  if (unit >= (int64_t)a.n_streams * groups) return;
  const int s = (int)(unit / groups), fg = (int)(unit % groups);
  FftUnitLds &L = U[wave];
  float *prod = L.prod();
  float *rv = L.small(), *normv = rv + 32, *gv = rv + 64;  // band vectors the per-bin interpolation gathers from
  const int f_end = (fg + 1) * kFramesPerWave < a.n_frames ? (fg + 1) * kFramesPerWave : a.n_frames;
  for (int f = fg * kFramesPerWave; f < f_end; ++f) {
    const int64_t cell = (int64_t)f * a.n_streams + s;
    const SuppFrameRec *rec = a.rec + cell;
    const float2 *Xg = a.X + cell * kRnnFreq;
    float2 *Pg = a.P + cell * kRnnFreq;
    float2 Xr[8], Pr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = lane + 64 * j;
      Xr[j] = i < kRnnFreq ? Xg[i] : make_float2(0.0f, 0.0f);
      Pr[j] = i < kRnnFreq ? Pg[i] : make_float2(0.0f, 0.0f);
    }
    float Ex = 0.0f, Ep = 0.0f, Exp = 0.0f, g = 0.0f, graw = 0.0f;
    if (lane < kRnnBands) {
      Ex = rec->Ex[lane];
      Ep = rec->Ep[lane];
      Exp = rec->Exp[lane];
      g = rec->gains[lane];
      graw = rec->gains_raw[lane];
    }
    const bool silence = rec->silence != 0;
    if (!silence) {
      // ---- pitch_filter (denoise.c): comb-filter the bands the network trusts less than the pitch
      if (lane < kRnnBands) {
        const float e = Exp;
        float r;
        if (e > graw) r = 1;
        else r = e * e * (1 - graw * graw) / (.001f + graw * graw * (1 - e * e));
        r = sqrtf(fminf(1.0f, fmaxf(0.0f, r)));
        r *= sqrtf(Ex / (1e-8f + Ep));
        rv[lane] = r;
        gv[lane] = g;
      }
      wave_lds_fence();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        if (i < kRnnFreq) {
          const float rf = interp_gain(rv, S, i);
          Xr[j].x += rf * Pr[j].x;
          Xr[j].y += rf * Pr[j].y;
          if (i < 400) prod[band_skew(i)] = Xr[j].x * Xr[j].x + Xr[j].y * Xr[j].y;
        }
      }
      {
        const float newE = band_sums_wave(prod, L.scratch(), S, lane);
        if (lane < kRnnBands) normv[lane] = sqrtf(Ex / (1e-8f + newE));
      }
      wave_lds_fence();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        if (i < kRnnFreq) {
          const float nf = interp_gain(normv, S, i);
          float2 v = Xr[j];
          v.x *= nf;
          v.y *= nf;
          const float gf = interp_gain(gv, S, i);  // band gains after the lastg floor
          v.x *= gf;
          v.y *= gf;
          Xr[j] = v;
        }
      }
      wave_lds_fence();  // every gather from the band vectors is done: the transform buffer may be filled
    }
    // ---- frame_synthesis: inverse transform through the forward FFT of the Hermitian extension
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = lane + 64 * j;
      if (i < kRnnFreq) {
        L.fa[i] = Xr[j];
        if (i > 0 && i < kRnnFrame) L.fa[kRnnWindow - i] = make_float2(Xr[j].x, -Xr[j].y);
      }
    }
    wave_lds_fence();
    fft960_wave(L.fa, S, lane, 1.0f);
    float *y = reinterpret_cast<float *>(Pg);
#pragma unroll
    for (int j = 0; j < 15; ++j) {
      const int i = lane + 64 * j;
      y[i] = L.fa[(kRnnWindow - i) % kRnnWindow].x * window_at(S, i);
    }
    wave_lds_fence();
  }
}

// Resynthesis and overlap-add as ONE kernel (round 3): a wave takes a stream through ALL frames of the window in order, so the
// 960 windowed samples of a frame never leave the CU -- the first half meets the previous frame's second half (eight values
// per lane, in registers; across windows: the stream's synthesis memory) and goes out as finished audio, the second half
// waits in registers for the next frame.  Against the two-kernel form this removes the write of every windowed frame over its
// P cell, its read (and the re-read of the previous frame's half) by supp_overlap_kernel, that kernel's launch and its
// one-wave-per-stream walk: 10.5 KB of HBM traffic per frame and stream become 1.9 KB (the output) + 1.9 KB (the dry signal,
// only when the mix needs it).  Arithmetic per sample is the two kernels' own, operation for operation.
extern "C" __global__ __launch_bounds__(64 * kFftWaves, 3) void supp_synth_kernel(SuppArgs a, SuppTables tb) {
  __shared__ FftShared S;
  __shared__ FftUnitLds U[kFftWaves];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  fft_shared_init(S, tb, tid, 64 * kFftWaves);
  __syncthreads();
  const int s = blockIdx.x * kFftWaves + wave;
  if (s >= a.n_streams) return;
  FftUnitLds &L = U[wave];
  float *prod = L.prod();
  float *rv = L.small(), *normv = rv + 32, *gv = rv + 64;  // band vectors the per-bin interpolation gathers from
  float *st = a.state + (int64_t)s * SuppState::kCount;
  float smoothed = st[SuppState::kSmoothedStrength];
  float prev[8];  // the previous frame's second half: sample 480 + lane + 64 j
#pragma unroll
  for (int j = 0; j < 8; ++j) prev[j] = (lane + 64 * j) < kRnnFrame ? st[SuppState::kSynthMem + lane + 64 * j] : 0.0f;
  const bool dry_from_out = a.front_clamp || a.front_dc;
  for (int f = 0; f < a.n_frames; ++f) {
    const int64_t cell = (int64_t)f * a.n_streams + s;
    const SuppFrameRec *rec = a.rec + cell;
    const float2 *Xg = a.X + cell * kRnnFreq;
    const float2 *Pg = a.P + cell * kRnnFreq;
    float2 Xr[8], Pr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = lane + 64 * j;
      Xr[j] = i < kRnnFreq ? Xg[i] : make_float2(0.0f, 0.0f);
      Pr[j] = i < kRnnFreq ? Pg[i] : make_float2(0.0f, 0.0f);
    }
    float Ex = 0.0f, Ep = 0.0f, Exp = 0.0f, g = 0.0f, graw = 0.0f;
    if (lane < kRnnBands) {
      Ex = rec->Ex[lane];
      Ep = rec->Ep[lane];
      Exp = rec->Exp[lane];
      g = rec->gains[lane];
      graw = rec->gains_raw[lane];
    }
    const bool silence = rec->silence != 0;
    // wet/dry smoothing, rnnoise.rs:81-86 (once per frame); the dry samples travel while the frame is transformed
    smoothed = a.strength * a.smoothing_coeff + smoothed * (1.0f - a.smoothing_coeff);
    const bool mix = !a.raw_protocol && smoothed < 1.0f;
    const int64_t base = (int64_t)s * a.stream_stride + (a.frame0 + f) * kRnnFrame;
    if (!silence) {
      // ---- pitch_filter (denoise.c): comb-filter the bands the network trusts less than the pitch
      if (lane < kRnnBands) {
        const float e = Exp;
        float r;
        if (e > graw) r = 1;
        else r = e * e * (1 - graw * graw) / (.001f + graw * graw * (1 - e * e));
        r = sqrtf(fminf(1.0f, fmaxf(0.0f, r)));
        r *= sqrtf(Ex / (1e-8f + Ep));
        rv[lane] = r;
        gv[lane] = g;
      }
      wave_lds_fence();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        if (i < kRnnFreq) {
          const float rf = interp_gain(rv, S, i);
          Xr[j].x += rf * Pr[j].x;
          Xr[j].y += rf * Pr[j].y;
          if (i < 400) prod[band_skew(i)] = Xr[j].x * Xr[j].x + Xr[j].y * Xr[j].y;
        }
      }
      {
        const float newE = band_sums_wave(prod, L.scratch(), S, lane);
        if (lane < kRnnBands) normv[lane] = sqrtf(Ex / (1e-8f + newE));
      }
      wave_lds_fence();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        if (i < kRnnFreq) {
          const float nf = interp_gain(normv, S, i);
          float2 v = Xr[j];
          v.x *= nf;
          v.y *= nf;
          const float gf = interp_gain(gv, S, i);  // band gains after the lastg floor
          v.x *= gf;
          v.y *= gf;
          Xr[j] = v;
        }
      }
      wave_lds_fence();  // every gather from the band vectors is done: the transform buffer may be filled
    }
    // ---- frame_synthesis: inverse transform through the forward FFT of the Hermitian extension
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = lane + 64 * j;
      if (i < kRnnFreq) {
        L.fa[i] = Xr[j];
        if (i > 0 && i < kRnnFrame) L.fa[kRnnWindow - i] = make_float2(Xr[j].x, -Xr[j].y);
      }
    }
    wave_lds_fence();
    fft960_wave(L.fa, S, lane, 1.0f);
    float dry[8];  // (fetched behind the transform: in front of it the eight registers spill)
    if (mix) {
      const float *dp = dry_from_out ? a.out + base : a.in + (int64_t)s * a.in_stride + (a.frame0 + f) * kRnnFrame;
#pragma unroll
      for (int j = 0; j < 8; ++j) dry[j] = (lane + 64 * j) < kRnnFrame ? dp[lane + 64 * j] : 0.0f;
    }
    // ---- overlap-add (first half + the previous frame's second half), /32768, wet/dry mix (rnnoise.rs:81-160)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = lane + 64 * j;
      if (i < kRnnFrame) {
        const float y_first = L.fa[(kRnnWindow - i) % kRnnWindow].x * window_at(S, i);
        const float y_second = L.fa[kRnnFrame - i].x * window_at(S, kRnnFrame + i);  // sample 480 + i sits at slot (960 - 480 - i)
        float wet = (y_first + prev[j]) / 32768.0f;
        if (mix) wet = (smoothed * wet) + ((1.0f - smoothed) * dry[j]);
        a.out[base + i] = wet;
        prev[j] = y_second;
      }
    }
    wave_lds_fence();
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if ((lane + 64 * j) < kRnnFrame) st[SuppState::kSynthMem + lane + 64 * j] = prev[j];
  if (lane == 0) st[SuppState::kSmoothedStrength] = smoothed;
}

// One wave per stream, frames in order: overlap-add of the windowed frames, /32768, smoothed wet/dry mix.
extern "C" __global__ __launch_bounds__(64) void supp_overlap_kernel(SuppArgs a) {
  const int lane = threadIdx.x;
  const int s = blockIdx.x;
  float *st = a.state + (int64_t)s * SuppState::kCount;
  float smoothed = st[SuppState::kSmoothedStrength];
  for (int f = 0; f < a.n_frames; ++f) {
    const float *y = reinterpret_cast<const float *>(a.P + ((int64_t)f * a.n_streams + s) * kRnnFreq);
    const float *prev = f == 0 ? st + SuppState::kSynthMem
                               : reinterpret_cast<const float *>(a.P + ((int64_t)(f - 1) * a.n_streams + s) * kRnnFreq) + kRnnFrame;
    // wet/dry smoothing, rnnoise.rs:81-86 (once per frame)
    smoothed = a.strength * a.smoothing_coeff + smoothed * (1.0f - a.smoothing_coeff);
    const int64_t base = (int64_t)s * a.stream_stride + (a.frame0 + f) * kRnnFrame;
#pragma unroll
    for (int i = lane; i < kRnnFrame; i += 64) {
      float wet = (y[i] + prev[i]) / 32768.0f;
      if (!a.raw_protocol && smoothed < 1.0f) {
        const float dry = (a.front_clamp || a.front_dc) ? a.out[base + i] : a.in[(int64_t)s * a.in_stride + (a.frame0 + f) * kRnnFrame + i];
        wet = (smoothed * wet) + ((1.0f - smoothed) * dry);
      }
      a.out[base + i] = wet;
    }
  }
  __syncthreads();  // frame 0 read the old synthesis memory above
  const float *last = reinterpret_cast<const float *>(a.P + ((int64_t)(a.n_frames - 1) * a.n_streams + s) * kRnnFreq) + kRnnFrame;
  for (int i = lane; i < kRnnFrame; i += 64) st[SuppState::kSynthMem + i] = last[i];
  if (lane == 0) st[SuppState::kSmoothedStrength] = smoothed;
}

// ============================================================================== launch
// The sample-serial pre-pass of a window (independent of the other kernels: it may run a window ahead).
template <bool kClamp, bool kDcHp, bool kRaw>
static hipError_t launch_prefilter_variant(const SuppArgs &a, hipStream_t stream) {
  constexpr size_t lds = sizeof(float) * 8 * kPreTile;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(supp_prefilter_kernel<kClamp, kDcHp, kRaw>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    attr_set = true;
  }
  const dim3 grid((a.n_streams + kPreGroup - 1) / kPreGroup), block(64 * kPreWaves);
  hipLaunchKernelGGL((supp_prefilter_kernel<kClamp, kDcHp, kRaw>), grid, block, lds, stream, a);
  return hipGetLastError();
}
hipError_t launch_suppressor_prefilter(const SuppArgs &a, hipStream_t stream) {
  const int sel = (a.front_clamp ? 4 : 0) | (a.front_dc ? 2 : 0) | (a.raw_protocol ? 1 : 0);
  switch (sel) {
    case 0: return launch_prefilter_variant<false, false, false>(a, stream);
    case 1: return launch_prefilter_variant<false, false, true>(a, stream);
    case 2: return launch_prefilter_variant<false, true, false>(a, stream);
    case 3: return launch_prefilter_variant<false, true, true>(a, stream);
    case 4: return launch_prefilter_variant<true, false, false>(a, stream);
    case 5: return launch_prefilter_variant<true, false, true>(a, stream);
    case 6: return launch_prefilter_variant<true, true, false>(a, stream);
    default: return launch_prefilter_variant<true, true, true>(a, stream);
  }
}

// Analysis of one window: spectra, then pitch + cepstral features (frames in order per stream).
// `before_pitch` (optional) is waited on between the spectra and the pitch search: the pitch search is 50 one-wave
// workgroups per stream with a small footprint, and while its grid drains, kernels with larger workgroups (the network:
// 256 VGPRs per wave) are not dispatched at all -- so the caller orders it after the previous window's network launch.
hipError_t launch_suppressor_analysis(const SuppArgs &a, const SuppTables &tb, hipStream_t stream, hipEvent_t before_pitch) {
  const int64_t units = (int64_t)a.n_streams * ((a.n_frames + kFramesPerWave - 1) / kFramesPerWave);
  const unsigned cells = (unsigned)((units + kFftWaves - 1) / kFftWaves);  // four units (waves) per transform workgroup
  hipLaunchKernelGGL(supp_spectrum_kernel, dim3(cells), dim3(64 * kFftWaves), 0, stream, a, tb);
  if (before_pitch) {
    hipError_t err = hipStreamWaitEvent(stream, before_pitch, 0);
    if (err != hipSuccess) return err;
  }
  static const bool search4 = [] {  // AF_PITCHSEARCH4=0: the round-2 form, one frame (wave) per workgroup (same-box A/B)
    const char *env = std::getenv("AF_PITCHSEARCH4");
    return !env || std::atoi(env) != 0;
  }();
  if (search4) {
    const unsigned groups = (unsigned)((a.n_frames + kPsFrames - 1) / kPsFrames);
    hipLaunchKernelGGL(supp_pitchsearch4_kernel, dim3((unsigned)a.n_streams * groups), dim3(64 * kPsFrames), 0, stream, a, tb);
  } else {
    hipLaunchKernelGGL(supp_pitchsearch_kernel, dim3((unsigned)((int64_t)a.n_streams * a.n_frames)), dim3(64), 0, stream, a, tb);
  }
  hipLaunchKernelGGL(supp_pitch_kernel, dim3(a.n_streams), dim3(64), 0, stream, a, tb);
  return hipGetLastError();
}

// The rest of the window: pitch-aligned spectra, the network, resynthesis, overlap-add.
// `after_network` (optional) is recorded right behind the network launch (see launch_suppressor_analysis).
// `finish_stream` (optional, needs `after_network`): resynthesis and overlap-add run there, so that the next window's
// pitch spectra and network launch can start while they run.
// `network_stream` + `after_spectra` (both or neither): the network kernel runs there, behind the pitch-spectrum kernel's event --
// the two are the longest pair of dependent kernels of a window, and one stream ran them back to back window after window.
hipError_t launch_suppressor_synthesis(const SuppArgs &a, const SuppTables &tb, const RnnDeviceWeights &w, hipStream_t stream,
                                       hipEvent_t after_network, hipStream_t finish_stream, hipStream_t network_stream,
                                       hipEvent_t after_spectra) {
  const int64_t units = (int64_t)a.n_streams * ((a.n_frames + kFramesPerWave - 1) / kFramesPerWave);
  const unsigned cells = (unsigned)((units + kFftWaves - 1) / kFftWaves);  // four units (waves) per transform workgroup
  hipLaunchKernelGGL(supp_pitchspec_kernel, dim3(cells), dim3(64 * kFftWaves), 0, stream, a, tb);
  if (network_stream && after_spectra && network_stream != stream) {
    hipError_t err = hipEventRecord(after_spectra, stream);
    if (err == hipSuccess) err = hipStreamWaitEvent(network_stream, after_spectra, 0);
    if (err != hipSuccess) return err;
    stream = network_stream;  // (what follows -- the network, its event -- is this stream's)
  }
  {
    // AF_RNN_VARIANT = waves (16 streams each) per workgroup: 1, 2 or 4.  Measured on one box, full bench step:
    // 4 -> 287-290 ms, 1 -> 298 ms (the round's first network kernel, a 4-wave workgroup per 16 streams: 301 ms)
    static const int variant = [] {
      const char *env = std::getenv("AF_RNN_VARIANT");
      return env ? std::atoi(env) : 4;
    }();
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(supp_rnn_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      if (err != hipSuccess) return err;
      attr_set = true;
    }
    const unsigned groups = (unsigned)((a.n_streams + 15) / 16);
    switch (variant) {
      case 1: hipLaunchKernelGGL(supp_rnn_kernel<1>, dim3(groups), dim3(64), rnn_lds_bytes(1), stream, a, w); break;
      case 2: hipLaunchKernelGGL(supp_rnn_kernel<2>, dim3((groups + 1) / 2), dim3(128), rnn_lds_bytes(2), stream, a, w); break;
      default: hipLaunchKernelGGL(supp_rnn_kernel<4>, dim3((groups + 3) / 4), dim3(256), rnn_lds_bytes(4), stream, a, w); break;
    }
  }
  if (after_network) {
    hipError_t err = hipEventRecord(after_network, stream);
    if (err != hipSuccess) return err;
  }
  hipStream_t fin = stream;
  if (finish_stream && finish_stream != stream && after_network) {
    hipError_t err = hipStreamWaitEvent(finish_stream, after_network, 0);
    if (err != hipSuccess) return err;
    fin = finish_stream;
  }
  static const bool fused_synthesis = [] {  // AF_SYNTH_FUSED=0: resynthesis and overlap-add as two kernels (round 2; same-box A/B)
    const char *env = std::getenv("AF_SYNTH_FUSED");
    return !env || std::atoi(env) != 0;
  }();
  if (fused_synthesis) {
    hipLaunchKernelGGL(supp_synth_kernel, dim3((unsigned)((a.n_streams + kFftWaves - 1) / kFftWaves)), dim3(64 * kFftWaves), 0, fin, a, tb);
  } else {
    hipLaunchKernelGGL(supp_resynth_kernel, dim3(cells), dim3(64 * kFftWaves), 0, fin, a, tb);
    hipLaunchKernelGGL(supp_overlap_kernel, dim3(a.n_streams), dim3(64), 0, fin, a);
  }
  return hipGetLastError();
}

}  // namespace af
