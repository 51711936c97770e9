// af_stages.hip -- Kernel 4: the dynamics chain (compressor -> limiter -> true-peak limiter -> output detector) as a
// PIPELINE OF STAGE KERNELS.
//
// The token-ring kernel (af_ring_kernel.hip) keeps a 64-stream group on one CU: whatever the batch, a launch lasts as long
// as one workgroup needs for its streams' samples (~1000 cycles per sample step), 256 streams use 4 CUs and 4096 use 64.
// The chain is a string of short recurrences (a few dependent f64 operations per sample each) separated by feed-forward
// math (log10 / exp10 / sqrt / divisions, two 128-tap FIRs, a sliding maximum).  Here every recurrence is a kernel of its
// own -- one wave per 64-stream group, lane = stream, its state in REGISTERS for the whole window, nothing but the
// recurrence in its loop -- and every feed-forward piece is a wide elementwise kernel over (sample, stream).  The kernels
// of a window run back to back on streams of their own, so while stage k works on window w stage k+1 works on window
// w-1: a launch set lasts as long as the SLOWEST STAGE needs per sample, not the sum, and the feed-forward work spreads
// over the whole chip.  Hand-over between stages is through time-major rings in HBM (af_stages.h): at 4096 streams that
// is ~0.2 KB per sample step per stream of extra traffic, two orders of magnitude under the HBM roof for this work.
//
// Arithmetic: the expressions are those of the token-ring kernel, operation for operation (the build does not contract
// floating-point expressions), so the two kernels agree bit for bit; tests/test_gpu_stages.py holds them to that.
//
// Not built in this form (the host keeps such configurations on kernel 2): the de-esser's EQ-first order, auto-makeup,
// a pending EQ crossfade, the time-major boundary layout.
#include <hip/hip_runtime.h>

#include "af_dsp.h"
#include "af_stages.h"

namespace af {
namespace {

constexpr int kU = 16;        // steps per unrolled block of a serial stage
constexpr int kTileRows = 64; // rows per workgroup of a feed-forward stage (4 waves x 16 rows)

__device__ __forceinline__ int64_t rrow(int64_t n, int rows) { return (n & (int64_t)(rows - 1)) * kLanes; }

// what every stage works out first
struct Who {
  int lane, g, s, sc;
  bool valid;
  int64_t NS;
};
__device__ __forceinline__ Who who(const StageArgs &a, int g) {
  Who w;
  w.lane = threadIdx.x & (kLanes - 1);
  w.g = g;
  w.s = g * kLanes + w.lane;
  w.valid = w.s < a.n_streams;
  w.sc = w.valid ? w.s : a.n_streams - 1;
  w.NS = a.n_streams;
  return w;
}
__device__ __forceinline__ const ChainParams &preset(const StageArgs &a, int g) {
  return a.params[a.group_preset ? a.group_preset[g] : 0];
}

// input of a serial stage: three blocks of kU rows in registers, the loads of block i + 3 issued when block i starts
template <typename T>
struct Ahead {
  T cur[kU], n1[kU], n2[kU];
  const T *base;  // ring of this group, at this lane
  int rows;
  int64_t n0;
  __device__ __forceinline__ void fill(T (&v)[kU], int64_t t) {
#pragma unroll
    for (int u = 0; u < kU; ++u) v[u] = base[rrow(n0 + t + u, rows)];
  }
  __device__ __forceinline__ void init(const T *ring, int g, int lane, int rows_, int64_t n0_, int64_t shift = 0) {
    base = ring + (int64_t)g * rows_ * kLanes + lane;
    rows = rows_;
    n0 = n0_ + shift;
    fill(cur, 0);
    fill(n1, kU);
    fill(n2, 2 * kU);
  }
  __device__ __forceinline__ void advance(int64_t t) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      cur[u] = n1[u];
      n1[u] = n2[u];
    }
    fill(n2, t + 3 * kU);
  }
};


// A block of kU steps of a serial stage that keeps per-control-block bookkeeping: the steps run in unrolled, unguarded
// form unless a control block (or the window) ends inside the kU steps; `block_end` exists once, outside the unrolled code.
template <typename Step, typename BlockEnd>
__device__ __forceinline__ void run_steps(int64_t t, int64_t n, int cb, int &in_block, Step step, BlockEnd block_end) {
  const int avail = (n - t) < kU ? (int)(n - t) : kU;
  int u0 = 0;
  while (u0 < avail) {
    int seg = avail - u0;
    if (cb - in_block < seg) seg = cb - in_block;
    if (seg == kU) {
#pragma unroll
      for (int u = 0; u < kU; ++u) step(u);
    } else {
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (u >= u0 && u < u0 + seg) step(u);
    }
    in_block += seg;
    u0 += seg;
    if (in_block == cb || t + u0 == n) {
      block_end();
      in_block = 0;
    }
  }
}

// ============================================================================================ stream-major -> time-major
__global__ __launch_bounds__(256) void stage_tin_kernel(StageArgs a) {
  __shared__ float tile[kTileRows][kLanes + 1];
  const int g = blockIdx.y;
  const Who w = who(a, g);
  const int wave = threadIdx.x >> 6;
  const int64_t t0 = (int64_t)blockIdx.x * kTileRows;
#pragma unroll 4
  for (int r = wave * 16; r < wave * 16 + 16; ++r) {
    const int s = g * kLanes + r;
    const int64_t t = t0 + w.lane;
    tile[w.lane][r] = (s < a.n_streams && t < a.n) ? a.in[(int64_t)s * a.stream_stride + t] : 0.0f;
  }
  __syncthreads();
  float *xe = a.r.xe + (int64_t)g * a.r.rows_f32 * kLanes + w.lane;
#pragma unroll 4
  for (int tt = wave * 16; tt < wave * 16 + 16; ++tt)
    if (t0 + tt < a.n) xe[rrow(a.n0 + t0 + tt, a.r.rows_f32)] = tile[tt][w.lane];
}

// ============================================================================================ compressor, serial part A
// side-chain high-pass + band / rms envelopes (compressor.rs:700-733; the token-ring kernel's token A)
__global__ __launch_bounds__(64) void stage_comp_a_kernel(StageArgs a) {
  const Who w = who(a, blockIdx.x);
  const ChainParams &P = preset(a, w.g);
  const CompressorParams &cp = P.comp;
  __builtin_amdgcn_s_setprio(3);
  Ahead<float> in;
  in.init(a.r.xe, w.g, w.lane, a.r.rows_f32, a.n0);
  const int R = a.r.rows_f64;
  const int64_t gb = (int64_t)w.g * R * kLanes + w.lane;
  double *o_d = a.r.d + gb, *o_low = a.r.low_e + gb, *o_voiced = a.r.voiced_e + gb, *o_pres = a.r.pres_e + gb, *o_rms = a.r.rms_e + gb;
  double rms_env = a.st64[(int64_t)kCompRmsEnvSq * w.NS + w.sc];
  double prev_in = a.st64[(int64_t)kCompScPrevIn * w.NS + w.sc], prev_out = a.st64[(int64_t)kCompScPrevOut * w.NS + w.sc];
  double low_env = a.st64[(int64_t)kCompLowEnv * w.NS + w.sc], voiced_env = a.st64[(int64_t)kCompVoicedEnv * w.NS + w.sc];
  double presence_env = a.st64[(int64_t)kCompPresenceEnv * w.NS + w.sc];
  // every parameter the loop reads is copied out first: read through the parameter pointer it would be re-loaded from
  // memory at every step (the loop's stores could alias it), each time behind a wait for ALL outstanding memory traffic
  const bool sc_on = cp.sidechain_highpass_enabled != 0;
  const double kk = cp.band_env_coeff, sc_coeff = cp.sidechain_highpass_coeff, rms_coeff = cp.rms_coeff;
  const int64_t n = a.n, n0 = a.n0;
  auto step = [&](int64_t t, float x) {
    const int64_t row = rrow(n0 + t, R);
    const double xin = (double)x;
    if (sc_on) {
      const double dd = sc_coeff * (prev_out + xin - prev_in);
      prev_in = xin;
      prev_out = dd;
      const double low = xin - dd;
      const double presence = 0.65 * dd + 0.35 * (dd - low);
      low_env = kk * low_env + (1.0 - kk) * low * low;
      voiced_env = kk * voiced_env + (1.0 - kk) * dd * dd;
      presence_env = kk * presence_env + (1.0 - kk) * presence * presence;
      rms_env = rms_coeff * rms_env + (1.0 - rms_coeff) * (dd * dd);
      o_d[row] = dd;
      o_low[row] = low_env;
      o_voiced[row] = voiced_env;
      o_pres[row] = presence_env;
      o_rms[row] = rms_env;
    } else {
      const double dd = xin;
      rms_env = rms_coeff * rms_env + (1.0 - rms_coeff) * (dd * dd);
      o_d[row] = dd;
      o_rms[row] = rms_env;
    }
  };
  for (int64_t t = 0; t < n; t += kU) {
    if (t + kU <= n) {
#pragma unroll
      for (int u = 0; u < kU; ++u) step(t + u, in.cur[u]);
    } else {
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (t + u < n) step(t + u, in.cur[u]);
    }
    in.advance(t);
  }
  if (w.valid) {
    a.st64[(int64_t)kCompRmsEnvSq * w.NS + w.s] = rms_env;
    if (sc_on) {
      a.st64[(int64_t)kCompScPrevIn * w.NS + w.s] = prev_in;
      a.st64[(int64_t)kCompScPrevOut * w.NS + w.s] = prev_out;
      a.st64[(int64_t)kCompLowEnv * w.NS + w.s] = low_env;
      a.st64[(int64_t)kCompVoicedEnv * w.NS + w.s] = voiced_env;
      a.st64[(int64_t)kCompPresenceEnv * w.NS + w.s] = presence_env;
    }
  }
}

// ============================================================================================ feed-forward 1
// detector weight, instantaneous peak and RMS levels in dB (update_sidechain_band_metrics, compressor.rs:438-449)
__global__ __launch_bounds__(256) void stage_f1_kernel(StageArgs a) {
  const int g = blockIdx.y;
  const Who w = who(a, g);
  const ChainParams &P = preset(a, g);
  const CompressorParams cp = P.comp;  // by value: fields read through the pointer would be re-loaded after every store
  const int wave = threadIdx.x >> 6;
  const int R = a.r.rows_f64;
  const int64_t gb = (int64_t)g * R * kLanes + w.lane;
  const int64_t t0 = (int64_t)blockIdx.x * kTileRows + wave * 16;
  const int64_t n = a.n;
  for (int k = 0; k < 16; ++k) {
    const int64_t t = t0 + k;
    if (t >= n) break;
    const int64_t row = gb + rrow(a.n0 + t, R);
    double weight_db = 0.0, plosive_last = 0.0;
    if (cp.sidechain_highpass_enabled) {
      const double low_rms = sqrt(a.r.low_e[row]);
      const double voiced_rms = fmax(sqrt(a.r.voiced_e[row]), 1e-8);
      const double presence_rms = sqrt(a.r.pres_e[row]);
      const double plosive = dclamp(low_rms / voiced_rms, 0.0, 32.0);
      plosive_last = plosive;
      const double plosive_amount = dclamp(div_known(plosive - 1.25, 3.75, 1.0 / 3.75), 0.0, 1.0);
      const double plosive_penalty = 1.0 - plosive_amount * (1.0 - 0.35);
      const double presence_ratio = dclamp(presence_rms / voiced_rms, 0.0, 4.0);
      const double presence_weight = 1.0 + 0.18 * dclamp(presence_ratio - 0.75, 0.0, 1.0);
      weight_db = lin2db(dclamp(plosive_penalty * presence_weight, 0.35, 1.15), 1e-10);
    }
    a.r.w_db[row] = weight_db;
    a.r.ipk_db[row] = lin2db(fabs(a.r.d[row]), 1e-10);
    a.r.rms_db[row] = lin2db(sqrt(a.r.rms_e[row]), 1e-10);
    if (t == n - 1 && w.valid) a.st64[(int64_t)kCompPlosive * w.NS + w.s] = plosive_last;  // diagnostic state only
  }
}

// ============================================================================================ compressor, serial part C
// log-domain peak envelope (compressor.rs:735-742)
__global__ __launch_bounds__(64) void stage_comp_c_kernel(StageArgs a) {
  const Who w = who(a, blockIdx.x);
  const ChainParams &P = preset(a, w.g);
  const CompressorParams &cp = P.comp;
  __builtin_amdgcn_s_setprio(3);
  Ahead<double> in;
  in.init(a.r.ipk_db, w.g, w.lane, a.r.rows_f64, a.n0);
  const int R = a.r.rows_f64;
  double *o = a.r.peak_db + (int64_t)w.g * R * kLanes + w.lane;
  double pe = a.st64[(int64_t)kCompPeakEnvDb * w.NS + w.sc];
  const double attack_coeff = cp.attack_coeff, detector_release_coeff = cp.detector_release_coeff;
  const int64_t n = a.n, n0 = a.n0;
  auto step = [&](int64_t t, double v) {
    const double pk = v > pe ? attack_coeff : detector_release_coeff;
    pe = pk * pe + (1.0 - pk) * v;
    o[rrow(n0 + t, R)] = pe;
  };
  for (int64_t t = 0; t < n; t += kU) {
    if (t + kU <= n) {
#pragma unroll
      for (int u = 0; u < kU; ++u) step(t + u, in.cur[u]);
    } else {
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (t + u < n) step(t + u, in.cur[u]);
    }
    in.advance(t);
  }
  if (w.valid) a.st64[(int64_t)kCompPeakEnvDb * w.NS + w.s] = pe;
}

// ============================================================================================ feed-forward 2
// blended detector level -> static gain-reduction target (compressor.rs:744-750,657-678)
__global__ __launch_bounds__(256) void stage_f2_kernel(StageArgs a) {
  const int g = blockIdx.y;
  const Who w = who(a, g);
  const ChainParams &P = preset(a, g);
  const CompressorParams cp = P.comp;  // by value: fields read through the pointer would be re-loaded after every store
  const int wave = threadIdx.x >> 6;
  const int R = a.r.rows_f64;
  const int64_t gb = (int64_t)g * R * kLanes + w.lane;
  const int64_t t0 = (int64_t)blockIdx.x * kTileRows + wave * 16;
  const int64_t n = a.n;
  for (int k = 0; k < 16; ++k) {
    const int64_t t = t0 + k;
    if (t >= n) break;
    const int64_t row = gb + rrow(a.n0 + t, R);
    const double blended = 0.6 * db2lin(a.r.peak_db[row]) + 0.4 * db2lin(a.r.rms_db[row]);
    a.r.target[row] = comp_gain_reduction(cp, lin2db(blended, 1e-10) + a.r.w_db[row]);
  }
}

// ============================================================================================ compressor, serial part E
// release-time meter + gain-reduction smoothing, makeup gain per control block (compressor.rs:452-505,604-617,752-764)
__global__ __launch_bounds__(64) void stage_comp_e_kernel(StageArgs a) {
  const Who w = who(a, blockIdx.x);
  const ChainParams &P = preset(a, w.g);
  const CompressorParams &cp = P.comp;
  __builtin_amdgcn_s_setprio(3);
  Ahead<double> in;
  in.init(a.r.target, w.g, w.lane, a.r.rows_f64, a.n0);
  const int R = a.r.rows_f64;
  double *o = a.r.gr + (int64_t)w.g * R * kLanes + w.lane;
  double gr = a.st64[(int64_t)kCompGr * w.NS + w.sc], fast = a.st64[(int64_t)kCompFastEnv * w.NS + w.sc];
  double slow = a.st64[(int64_t)kCompSlowEnv * w.NS + w.sc];
  double cur_ms = a.st64[(int64_t)kCompCurReleaseMs * w.NS + w.sc], tgt_ms = a.st64[(int64_t)kCompTargetReleaseMs * w.NS + w.sc];
  const double rel_coeff = a.st64[(int64_t)kCompReleaseCoeff * w.NS + w.sc];
  double sm = a.st64[(int64_t)kCompSmoothedMakeup * w.NS + w.sc];
  double makeup_lin = db2lin(sm);
  const int cb = P.control_block;
  const bool adaptive = cp.adaptive_release != 0;
  const double base_release_ms = cp.base_release_ms, release_smoothing_coeff = cp.release_smoothing_coeff;
  const double attack_coeff = cp.attack_coeff, fast_release_coeff = cp.fast_release_coeff, slow_charge_coeff = cp.slow_charge_coeff;
  const double slow_release_coeff = cp.slow_release_coeff, makeup_smoothing_coeff = cp.makeup_smoothing_coeff, makeup_gain_db = cp.makeup_gain_db;
  const double sample_rate = cp.sample_rate;
  const int64_t n = a.n, n0 = a.n0;
  double *mk = a.mk;
  BlockStats *stats = a.stats;
  int in_block = 0;
  int64_t b = 0;
  if (w.valid) mk[w.s] = makeup_lin;  // the gain in force during the window's first block
  auto step = [&](int64_t t, double tg) {
    if (adaptive) {
      const double sustained = dclamp(div_known(slow, 6.0, 1.0 / 6.0), 0.0, 1.0);
      const double transient_bias = dclamp(div_known(fast - slow, 7.0, 1.0 / 7.0), 0.0, 1.0);
      const double syllabic = dclamp(sustained * sustained * (1.0 - 0.35 * transient_bias), 0.0, 1.0);
      tgt_ms = 50.0 + syllabic * (400.0 - 50.0);
    } else {
      tgt_ms = base_release_ms;
    }
    if (fabs(tgt_ms - cur_ms) > 1.0) {
      cur_ms = release_smoothing_coeff * cur_ms + (1.0 - release_smoothing_coeff) * tgt_ms;
    } else {
      cur_ms = tgt_ms;
    }
    if (!adaptive) {
      const double kk = tg > gr ? attack_coeff : rel_coeff;
      gr = kk * gr + (1.0 - kk) * tg;
      fast = gr;
      slow = 0.0;
    } else {
      if (tg > gr) {
        fast = attack_coeff * gr + (1.0 - attack_coeff) * tg;
      } else {
        fast = fast_release_coeff * fast + (1.0 - fast_release_coeff) * tg;
      }
      if (tg > 3.0) {
        slow = slow_charge_coeff * slow + (1.0 - slow_charge_coeff) * tg;
      } else {
        slow *= slow_release_coeff;
      }
      gr = fmax(fast, slow);
    }
    o[rrow(n0 + t, R)] = gr;
  };
  auto block_end = [&]() {  // wave-uniform: a control block (or the window) ends after the step just made
    const int blk_len = in_block;
    if (w.valid && stats) stats[b * w.NS + w.s].compressor_gr_db = (float)gr;
    // update_auto_makeup_gain with auto-makeup off (compressor.rs:604-617)
    const double makeup_coeff = pow(makeup_smoothing_coeff, (double)(blk_len < 1 ? 1 : blk_len));
    const double tgt = makeup_gain_db;
    if (fabs(tgt - sm) > 0.1) {
      sm = makeup_coeff * sm + (1.0 - makeup_coeff) * tgt;
    } else {
      sm = tgt;
    }
    makeup_lin = db2lin(sm);
    if (w.valid && stats) stats[b * w.NS + w.s].makeup_gain_db = (float)sm;
    b += 1;
    if (w.valid) mk[b * w.NS + w.s] = makeup_lin;  // ... and during the next one
  };
  for (int64_t t = 0; t < n; t += kU) {
    run_steps(t, n, cb, in_block, [&](int u) { step(t + u, in.cur[u]); }, block_end);
    in.advance(t);
  }
  if (w.valid) {
    a.st64[(int64_t)kCompGr * w.NS + w.s] = gr;
    a.st64[(int64_t)kCompFastEnv * w.NS + w.s] = fast;
    a.st64[(int64_t)kCompSlowEnv * w.NS + w.s] = slow;
    a.st64[(int64_t)kCompCurReleaseMs * w.NS + w.s] = cur_ms;
    a.st64[(int64_t)kCompTargetReleaseMs * w.NS + w.s] = tgt_ms;
    a.st64[(int64_t)kCompSmoothedMakeup * w.NS + w.s] = sm;
    if (adaptive) {
      const double tau = fmax(cur_ms, 0.001) / 1000.0;  // compressor.rs:760-761
      a.st64[(int64_t)kCompReleaseCoeff * w.NS + w.s] = exp(-1.0 / (tau * sample_rate));
    }
  }
}

// ============================================================================================ feed-forward 3
// apply gain (compressor.rs:771-773)
__global__ __launch_bounds__(256) void stage_f3_kernel(StageArgs a) {
  const int g = blockIdx.y;
  const Who w = who(a, g);
  const ChainParams &P = preset(a, g);
  const int wave = threadIdx.x >> 6;
  const int R = a.r.rows_f64, R32 = a.r.rows_f32;
  const int64_t gb = (int64_t)g * R * kLanes + w.lane, gb32 = (int64_t)g * R32 * kLanes + w.lane;
  const int64_t t0 = (int64_t)blockIdx.x * kTileRows + wave * 16;
  const int cb = P.control_block;
  for (int k = 0; k < 16; ++k) {
    const int64_t t = t0 + k;
    if (t >= a.n) break;
    const double makeup_lin = a.mk[(t / cb) * w.NS + w.sc];
    const double gr = a.r.gr[gb + rrow(a.n0 + t, R)];
    const int64_t row = gb32 + rrow(a.n0 + t, R32);
    a.r.xc[row] = (float)((double)a.r.xe[row] * (db2lin(-gr) * makeup_lin));
  }
}

// ============================================================================================ feed-forward 4
// lookahead limiter, the part without memory (limiter.rs:246-270): the maximum of |x| over the last W = lookahead + 1
// samples, from suffix maxima of the previous W-aligned block and the running prefix maximum of the current one (the
// maximum is exact whatever the grouping), and the gain it asks for.  One wave per (block that meets the window, group).
__global__ __launch_bounds__(64) void stage_f4_kernel(StageArgs a, const float *xin_ring) {
  const int g = blockIdx.y;
  const Who w = who(a, g);
  const ChainParams &P = preset(a, g);
  const int W = P.lim.lookahead_samples + 1;
  const double ceil_lin = P.lim.ceiling_linear;
  const int R32 = a.r.rows_f32, R = a.r.rows_f64;
  const float *x = xin_ring + (int64_t)g * R32 * kLanes + w.lane;
  float *sfx = a.r.sfx + (int64_t)g * R32 * kLanes + w.lane;
  double *tg = a.r.tg + (int64_t)g * R * kLanes + w.lane;
  // blocks are aligned to absolute multiples of W
  const int64_t first = a.n0 / W;
  const int64_t B = first + blockIdx.x;
  const int64_t b0 = B * W;
  if (b0 >= a.n0 + a.n) return;
  // suffix maxima of block B - 1, eight loads at a time
  {
    float m = 0.0f;
    int j = W - 1;
    for (; j >= 7; j -= 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = fabsf(x[rrow(b0 - W + j - u, R32)]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        m = fmaxf(m, v[u]);
        sfx[rrow(b0 - W + j - u, R32)] = m;
      }
    }
    for (; j >= 0; --j) {
      m = fmaxf(m, fabsf(x[rrow(b0 - W + j, R32)]));
      sfx[rrow(b0 - W + j, R32)] = m;
    }
  }
  // forward over block B: running prefix maximum, output for the rows that belong to this window
  float prefix = 0.0f;
  const int64_t end = (b0 + W < a.n0 + a.n) ? b0 + W : a.n0 + a.n;
  int64_t n = b0;
  for (; n + 8 <= end; n += 8) {
    float v[8], sf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      v[u] = fabsf(x[rrow(n + u, R32)]);
      const int64_t j = n + u - b0;
      sf[u] = (j + 1 < W) ? sfx[rrow(b0 - W + j + 1, R32)] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      prefix = (n + u == b0) ? v[u] : fmaxf(prefix, v[u]);
      if (n + u >= a.n0) {
        const double peak = (double)fmaxf(sf[u], prefix);
        tg[rrow(n + u, R)] = peak > ceil_lin ? ceil_lin / peak : 1.0;
      }
    }
  }
  for (; n < end; ++n) {
    const float ax = fabsf(x[rrow(n, R32)]);
    const int64_t j = n - b0;
    const float sf = (j + 1 < W) ? sfx[rrow(b0 - W + j + 1, R32)] : 0.0f;
    prefix = (n == b0) ? ax : fmaxf(prefix, ax);
    if (n >= a.n0) {
      const double peak = (double)fmaxf(sf, prefix);
      tg[rrow(n, R)] = peak > ceil_lin ? ceil_lin / peak : 1.0;
    }
  }
}

// ============================================================================================ limiter, serial part
// gain smoothing (limiter.rs:271-284)
__global__ __launch_bounds__(64) void stage_lim_kernel(StageArgs a) {
  const Who w = who(a, blockIdx.x);
  const ChainParams &P = preset(a, w.g);
  __builtin_amdgcn_s_setprio(3);
  Ahead<double> in;
  in.init(a.r.tg, w.g, w.lane, a.r.rows_f64, a.n0);
  const int R = a.r.rows_f64;
  double *o = a.r.g + (int64_t)w.g * R * kLanes + w.lane;
  const double rc = P.lim.release_coeff;
  double g = a.st64[(int64_t)kLimGain * w.NS + w.sc];
  double gmin = 1.0;
  const int cb = P.control_block;
  const int64_t n = a.n, n0 = a.n0;
  BlockStats *stats = a.stats;
  int in_block = 0;
  int64_t b = 0;
  auto step = [&](int64_t t, double tg) {
    if (tg < g) {
      g = tg;
    } else {
      g = rc * g + (1.0 - rc) * tg;
    }
    gmin = fmin(gmin, g);
    o[rrow(n0 + t, R)] = g;
  };
  auto block_end = [&]() {
    if (w.valid && stats) stats[b * w.NS + w.s].limiter_peak_gr_db = gmin < 1.0 ? (float)(-lin2db(gmin, 1e-10)) : 0.0f;
    gmin = 1.0;
    b += 1;
  };
  for (int64_t t = 0; t < n; t += kU) {
    run_steps(t, n, cb, in_block, [&](int u) { step(t + u, in.cur[u]); }, block_end);
    in.advance(t);
  }
  if (w.valid) a.st64[(int64_t)kLimGain * w.NS + w.s] = g;
}

// ============================================================================================ feed-forward 5
// limiter output (limiter.rs:278-284), input-side 4x true peak (true_peak.rs:173-186,341-352) and the gain it asks for.
// A workgroup owns 64 rows of a group; the limiter output of those rows and of the 31 before them goes through LDS.
__device__ __forceinline__ float tp_observe_regs(const float (&h)[kTpTaps + 15], int i) {
  // the window of row i: h[i + 31 - k] is the sample k steps back
  float peak = fabsf(h[i + kTpTaps - 1]);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < kTpTaps; ++k) acc = __builtin_fmaf(AF_TP_FIR[p][k], h[i + kTpTaps - 1 - k], acc);
    peak = fmaxf(peak, fabsf(acc));
  }
  return peak;
}

__global__ __launch_bounds__(256) void stage_f5_kernel(StageArgs a, const float *xin_ring) {
  __shared__ float xl_t[kTileRows + kTpTaps - 1][kLanes];
  const int g = blockIdx.y;
  const Who w = who(a, g);
  const ChainParams &P = preset(a, g);
  const int wave = threadIdx.x >> 6;
  const int R32 = a.r.rows_f32, R = a.r.rows_f64;
  const int64_t gb32 = (int64_t)g * R32 * kLanes + w.lane, gb = (int64_t)g * R * kLanes + w.lane;
  const int la = P.lim.lookahead_samples;
  const double ceil_lin = P.lim.ceiling_linear;
  const float tp_ceiling = P.tp.ceiling_linear;
  const int64_t t0 = (int64_t)blockIdx.x * kTileRows;  // window-relative first row of the tile
  // rows t0 - 31 .. t0 + 63 of the limiter output (rows before the stream's first sample: the rings hold zeros)
  for (int i = wave; i < kTileRows + kTpTaps - 1; i += 4) {
    const int64_t n = a.n0 + t0 - (kTpTaps - 1) + i;
    const float delayed = xin_ring[gb32 + rrow(n - la, R32)];
    const double gain = a.r.g[gb + rrow(n, R)];
    const float o = (float)dclamp((double)delayed * gain, -ceil_lin, ceil_lin);
    const float v = finite_f32(o) ? o : 0.0f;  // TruePeakLimiter input scrub, true_peak.rs:342
    xl_t[i][w.lane] = v;
    if (i >= kTpTaps - 1 && t0 + i - (kTpTaps - 1) < a.n) a.r.xl[gb32 + rrow(n, R32)] = v;
  }
  __syncthreads();
  float h[kTpTaps + 15];
#pragma unroll
  for (int i = 0; i < kTpTaps + 15; ++i) h[i] = xl_t[wave * 16 + i][w.lane];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t t = t0 + wave * 16 + k;
    if (t < a.n) {
      const float itp = tp_observe_regs(h, k);
      float tg = 1.0f;
      if (itp > tp_ceiling) tg = fclamp((tp_ceiling * 0.999f) / itp, 0.0f, 1.0f);
      a.r.itp[gb32 + rrow(a.n0 + t, R32)] = itp;
      a.r.tgt[gb32 + rrow(a.n0 + t, R32)] = tg;
    }
  }
}

// ============================================================================================ true-peak limiter, serial part
// gain (true_peak.rs:353-374), chain output, block output statistics (block_processor.rs:150-170)
template <bool kLim>
__global__ __launch_bounds__(64) void stage_tp_kernel(StageArgs a, const float *xin_ring) {
  const Who w = who(a, blockIdx.x);
  const ChainParams &P = preset(a, w.g);
  __builtin_amdgcn_s_setprio(3);
  const int R32 = a.r.rows_f32;
  Ahead<float> in_x, in_itp, in_tgt;
  if (kLim) {
    in_x.init(a.r.xl, w.g, w.lane, R32, a.n0, -kTpDelay);
    in_itp.init(a.r.itp, w.g, w.lane, R32, a.n0);
    in_tgt.init(a.r.tgt, w.g, w.lane, R32, a.n0);
  } else {
    in_x.init(xin_ring, w.g, w.lane, R32, a.n0);
  }
  float *o_ring = a.r.od + (int64_t)w.g * R32 * kLanes + w.lane;
  const float tp_ceiling = P.tp.ceiling_linear;
  const float rel = P.tp.release_coeff;
  const bool comp_on = (P.flags & kFlagCompressor) != 0;
  float g = kLim ? a.st32[(int64_t)kTpGain * w.NS + w.sc] : 1.0f;
  double out_sq = 0.0;
  float out_peak = 0.0f, nonfinite = 0.0f, tp_in_peak = 0.0f, tp_gmin = 1.0f, tp_limited = 0.0f;
  const int cb = P.control_block;
  const int64_t n = a.n, n0 = a.n0;
  BlockStats *stats = a.stats;
  int in_block = 0;
  int64_t b = 0;
  auto step = [&](int64_t t, float delayed, float itp, float tg) {
    float o = delayed;
    if (kLim) {
      tp_in_peak = fmaxf(tp_in_peak, itp);
      if (tg < g) {
        g = tg;
        tp_limited = 1.0f;
      } else {
        g = rel * g + (1.0f - rel) * tg;
      }
      tp_gmin = fminf(tp_gmin, g);
      o = fclamp(delayed * g, -tp_ceiling, tp_ceiling);
      if (!finite_f32(o)) o = 0.0f;
    }
    if (finite_f32(o)) {
      out_sq += (double)o * (double)o;
    } else {
      nonfinite = 1.0f;
    }
    out_peak = fmaxf(out_peak, fabsf(o));
    o_ring[rrow(n0 + t, R32)] = o;
  };
  auto block_end = [&]() {
    if (w.valid && stats) {
      BlockStats &row = stats[b * w.NS + w.s];
      row.output_square_sum = out_sq;
      row.output_sample_peak = out_peak;
      row.non_finite_output = nonfinite != 0.0f ? 1u : 0u;
      row.tp_limiter_input_peak = tp_in_peak;
      row.tp_limiter_gr_db = kLim && tp_gmin < 1.0f ? -20.0f * log10f(fmaxf(tp_gmin, 1e-10f)) : 0.0f;
      row.tp_limited_events = tp_limited != 0.0f ? 1u : 0u;
    }
    out_sq = 0.0;
    out_peak = 0.0f;
    nonfinite = 0.0f;
    tp_in_peak = 0.0f;
    tp_gmin = 1.0f;
    tp_limited = 0.0f;
    b += 1;
  };
  for (int64_t t = 0; t < n; t += kU) {
    run_steps(t, n, cb, in_block,
              [&](int u) { step(t + u, in_x.cur[u], kLim ? in_itp.cur[u] : 0.0f, kLim ? in_tgt.cur[u] : 1.0f); }, block_end);
    in_x.advance(t);
    if (kLim) {
      in_itp.advance(t);
      in_tgt.advance(t);
    }
  }
  if (w.valid) {
    if (kLim) a.st32[(int64_t)kTpGain * w.NS + w.s] = g;
    if (!comp_on) a.st64[(int64_t)kCompGr * w.NS + w.s] = 0.0;
  }
}

// ============================================================================================ feed-forward 6
// output-side 4x true peak (TruePeakDetector::process_block, true_peak.rs:205-221; block_processor.rs:159) folded into the
// block maximum, and the chain output back in stream-major order
__global__ __launch_bounds__(256) void stage_f6_kernel(StageArgs a) {
  __shared__ float tile[kTileRows][kLanes + 1];
  const int g = blockIdx.y;
  const Who w = who(a, g);
  const ChainParams &P = preset(a, g);
  const int wave = threadIdx.x >> 6;
  const int R32 = a.r.rows_f32;
  const float *od = a.r.od + (int64_t)g * R32 * kLanes + w.lane;
  const int64_t t0 = (int64_t)blockIdx.x * kTileRows;
  const int64_t tw = t0 + wave * 16;  // first row of this wave
  float h[kTpTaps + 15];
#pragma unroll
  for (int i = 0; i < kTpTaps + 15; ++i) {
    const float v = od[rrow(a.n0 + tw - (kTpTaps - 1) + i, R32)];
    h[i] = v;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) tile[wave * 16 + k][w.lane] = h[kTpTaps - 1 + k];
#pragma unroll
  for (int i = 0; i < kTpTaps + 15; ++i)
    if (!finite_f32(h[i])) h[i] = 0.0f;  // the detector scrubs what it is fed (true_peak.rs:212)
  const int cb = P.control_block;
  float m = 0.0f;
  int64_t mb = tw / cb;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t t = tw + k;
    if (t < a.n) {
      const int64_t b = t / cb;
      if (b != mb) {
        if (w.valid && a.stats) atomicMax(reinterpret_cast<unsigned int *>(&a.stats[mb * w.NS + w.s].output_true_peak), __float_as_uint(m));
        m = 0.0f;
        mb = b;
      }
      m = fmaxf(m, tp_observe_regs(h, k));
    }
  }
  if (tw < a.n && w.valid && a.stats)
    atomicMax(reinterpret_cast<unsigned int *>(&a.stats[mb * w.NS + w.s].output_true_peak), __float_as_uint(m));
  __syncthreads();
#pragma unroll 4
  for (int r = wave * 16; r < wave * 16 + 16; ++r) {
    const int s = g * kLanes + r;
    const int64_t t = t0 + w.lane;
    if (s < a.n_streams && t < a.n) a.out[(int64_t)s * a.stream_stride + t] = tile[w.lane][r];
  }
}

}  // namespace

// `flags`: the preset-0 chain flags (what the pipeline was planned for)
hipError_t launch_stage(int stage, const StageArgs &a, uint32_t flags, hipStream_t stream) {
  const int groups = (a.n_streams + kLanes - 1) / kLanes;
  const unsigned tiles = (unsigned)((a.n + kTileRows - 1) / kTileRows);
  const bool comp = (flags & kFlagCompressor) != 0;
  const float *lim_in = comp ? a.r.xc : a.r.xe;  // what the limiter (or, without one, the output stage) reads
  if (a.n <= 0) return hipSuccess;
  switch (stage) {
    case kStTin: hipLaunchKernelGGL(stage_tin_kernel, dim3(tiles, groups), dim3(256), 0, stream, a); break;
    case kStCompA: hipLaunchKernelGGL(stage_comp_a_kernel, dim3(groups), dim3(64), 0, stream, a); break;
    case kStF1: hipLaunchKernelGGL(stage_f1_kernel, dim3(tiles, groups), dim3(256), 0, stream, a); break;
    case kStCompC: hipLaunchKernelGGL(stage_comp_c_kernel, dim3(groups), dim3(64), 0, stream, a); break;
    case kStF2: hipLaunchKernelGGL(stage_f2_kernel, dim3(tiles, groups), dim3(256), 0, stream, a); break;
    case kStCompE: hipLaunchKernelGGL(stage_comp_e_kernel, dim3(groups), dim3(64), 0, stream, a); break;
    case kStF3: hipLaunchKernelGGL(stage_f3_kernel, dim3(tiles, groups), dim3(256), 0, stream, a); break;
    case kStF4: {
      // W-aligned blocks that meet [n0, n0 + n): at most n / W_min + 2; a wave whose block starts past the window returns
      hipLaunchKernelGGL(stage_f4_kernel, dim3((unsigned)(a.n / (a.w_min > 0 ? a.w_min : 1) + 2), groups), dim3(64), 0, stream, a, lim_in);
      break;
    }
    case kStLim: hipLaunchKernelGGL(stage_lim_kernel, dim3(groups), dim3(64), 0, stream, a); break;
    case kStF5: hipLaunchKernelGGL(stage_f5_kernel, dim3(tiles, groups), dim3(256), 0, stream, a, lim_in); break;
    case kStTp:
      if (flags & kFlagLimiter) hipLaunchKernelGGL(stage_tp_kernel<true>, dim3(groups), dim3(64), 0, stream, a, lim_in);
      else hipLaunchKernelGGL(stage_tp_kernel<false>, dim3(groups), dim3(64), 0, stream, a, lim_in);
      break;
    case kStF6: hipLaunchKernelGGL(stage_f6_kernel, dim3(tiles, groups), dim3(256), 0, stream, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace af
