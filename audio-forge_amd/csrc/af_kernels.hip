// af_kernels.hip -- hand-written gfx950 kernels for the batched voice chain.
//
// Kernel 1: chain_lane_kernel  ("lane per stream")
//   One wavefront (64 lanes) owns 64 consecutive streams for the whole launch.  Audio
//   moves HBM -> LDS in 64-sample x 64-stream tiles (each stream row is a coalesced 256 B
//   read; the tile is transposed on the way into LDS so that afterwards lane i walks
//   column i, conflict free).  Stages then run stage-major over the LDS tile exactly in
//   the reference's order -- [DC block + 80 Hz HP] -> 10-band EQ (band-major over the
//   tile, like eq.rs:371-379) -> compressor -> lookahead limiter -> 4x true-peak limiter
//   -> true-peak detector -- every stream strictly sample-sequential, every f64/f32
//   rounding point where the Rust text has it (-ffp-contract=off; explicit fmaf only in
//   the true-peak FIR).  It is the parity anchor for the faster kernels.
//
// Arithmetic follows rust-core/src/dsp/{biquad,eq,compressor,limiter,true_peak}.rs and
// audio/processor/{routing,block_processor}.rs; line numbers are cited per function.
#include <hip/hip_runtime.h>

#include "af_dsp.h"

namespace af {

struct TileLds {
  float x[kTile][kLanes + 1];            // working tile; +1 column: conflict-free transposed fill
  float tpi[kTpTaps + kTile][kLanes];    // true-peak limiter input  (rows 0..31 = history)
  float tpo[kTpTaps + kTile][kLanes];    // chain output             (rows 0..31 = history)
};

// ------------------------------------------------------------------ tile movement
__device__ __forceinline__ void load_tile(const LaunchArgs &a, TileLds &L, int64_t t0, int len, int s0,
                                          int lane) {
  if (a.layout == 0) {
    // stream-major: row r of the group is contiguous in time -> lane j reads sample t0+j
    for (int r = 0; r < kLanes; ++r) {
      const int s = s0 + r;
      float v = 0.0f;
      if (s < a.n_streams && lane < len) v = a.in[(int64_t)s * a.stream_stride + t0 + lane];
      L.x[lane][r] = v;
    }
  } else {
    const int s = s0 + lane;
    for (int t = 0; t < len; ++t) {
      float v = 0.0f;
      if (s < a.n_streams) v = a.in[(t0 + t) * a.stream_stride + s];
      L.x[t][lane] = v;
    }
  }
}

__device__ __forceinline__ void store_tile(const LaunchArgs &a, TileLds &L, int64_t t0, int len, int s0,
                                           int lane) {
  if (a.layout == 0) {
    for (int r = 0; r < kLanes; ++r) {
      const int s = s0 + r;
      if (s < a.n_streams && lane < len) a.out[(int64_t)s * a.stream_stride + t0 + lane] = L.x[lane][r];
    }
  } else {
    const int s = s0 + lane;
    if (s < a.n_streams)
      for (int t = 0; t < len; ++t) a.out[(t0 + t) * a.stream_stride + s] = L.x[t][lane];
  }
}

// ------------------------------------------------------------------ biquad section
// Biquad::process_sample, dsp/biquad.rs:263-327, over one tile column.
// `rem` (crossfade samples left) is uniform across lanes.
__device__ __forceinline__ void biquad_tile(float (*x)[kLanes + 1], int lane, int len, const SectionParams &sp,
                                            int rem, double &z1, double &z2, double &pz1, double &pz2) {
  BiquadCoef c = sp.active;
  int t = 0;
  if (rem > 0) {
    const BiquadCoef p = sp.pending;
    const double total = (double)sp.xf_total;
    for (; t < len && rem > 0; ++t) {
      const double in = (double)x[t][lane];
      const double ya = c.b0 * in + z1;
      z1 = c.b1 * in - c.a1 * ya + z2;
      z2 = c.b2 * in - c.a2 * ya;
      const double yp = p.b0 * in + pz1;
      pz1 = p.b1 * in - p.a1 * yp + pz2;
      pz2 = p.b2 * in - p.a2 * yp;
      const int fade_pos = sp.xf_total - rem + 1;
      const double fade = (double)fade_pos / total;
      const double y = ya * (1.0 - fade) + yp * fade;
      rem -= 1;
      if (rem == 0) {  // promote_pending_coefficients, biquad.rs:276-286
        c = p;
        z1 = pz1;
        z2 = pz2;
      }
      x[t][lane] = (float)y;
    }
  } else if (sp.xf_remaining > 0) {
    c = sp.pending;  // crossfade finished in an earlier tile of this launch
  }
  for (; t < len; ++t) {
    const double in = (double)x[t][lane];
    const double y = c.b0 * in + z1;
    z1 = c.b1 * in - c.a1 * y + z2;
    z2 = c.b2 * in - c.a2 * y;
    x[t][lane] = (float)y;
  }
}

// ------------------------------------------------------------------ compressor state
struct CompState {
  double sc_prev_in, sc_prev_out, low_env, voiced_env, presence_env, plosive;
  double peak_env_db, rms_env_sq, gr, fast_env, slow_env, cur_release_ms, target_release_ms;
  double release_coeff, smoothed_makeup;
};

// Compressor::process_sample_impl(update_makeup_gain = false), dsp/compressor.rs:725-774
__device__ __forceinline__ float comp_sample(const CompressorParams &p, CompState &s, float input,
                                             double makeup_lin) {
  const double x = (double)input;
  double d = x;
  double weight_db = kDetectorUnitWeight;  // (dB in the literal build, a linear factor otherwise: af_dsp.h, detector_db)
  if (p.sidechain_highpass_enabled) {
    // process_sidechain_sample, compressor.rs:407-417
    d = p.sidechain_highpass_coeff * (s.sc_prev_out + x - s.sc_prev_in);
    s.sc_prev_in = x;
    s.sc_prev_out = d;
    // update_sidechain_band_metrics, compressor.rs:420-450
    const double low = x - d;
    const double voiced = d;
    const double presence = 0.65 * d + 0.35 * (d - low);
    const double k = p.band_env_coeff;
    s.low_env = k * s.low_env + (1.0 - k) * low * low;
    s.voiced_env = k * s.voiced_env + (1.0 - k) * voiced * voiced;
    s.presence_env = k * s.presence_env + (1.0 - k) * presence * presence;
    const double low_rms = sqrt(s.low_env);
    const double voiced_rms = fmax(sqrt(s.voiced_env), 1e-8);
    const double presence_rms = sqrt(s.presence_env);
    s.plosive = dclamp(low_rms / voiced_rms, 0.0, 32.0);
    const double plosive_amount = dclamp(div_known(s.plosive - 1.25, 3.75, 1.0 / 3.75), 0.0, 1.0);
    const double plosive_penalty = 1.0 - plosive_amount * (1.0 - 0.35);
    const double presence_ratio = dclamp(presence_rms / voiced_rms, 0.0, 4.0);
    const double presence_weight = 1.0 + 0.18 * dclamp(presence_ratio - 0.75, 0.0, 1.0);
    const double w = dclamp(plosive_penalty * presence_weight, 0.35, 1.15);
    weight_db = detector_weight(w);
  } else {
    s.plosive = 0.0;
  }
  const double inst_peak_db = lin2db(fabs(d), 1e-10);
  const double pk = inst_peak_db > s.peak_env_db ? p.attack_coeff : p.detector_release_coeff;
  s.peak_env_db = pk * s.peak_env_db + (1.0 - pk) * inst_peak_db;

  const double sq = d * d;
  s.rms_env_sq = p.rms_coeff * s.rms_env_sq + (1.0 - p.rms_coeff) * sq;
  const double rms_db = detector_rms_level(s.rms_env_sq);

  // blended_detector_db, compressor.rs:681-686
  const double detector_db = af::detector_db(s.peak_env_db, rms_db, weight_db);

  // update_adaptive_release_time_meter + release smoothing, compressor.rs:452-466,752-761.
  // release_coeff = tc(current_release_ms) is only consumed by the non-adaptive branch, where
  // current_release_ms never moves off base_release_ms, so the host-computed value is exact;
  // in adaptive mode the per-sample exp() result is dead and is refreshed at launch end.
  if (p.adaptive_release) {
    const double sustained = dclamp(div_known(s.slow_env, 6.0, 1.0 / 6.0), 0.0, 1.0);
    const double transient_bias = dclamp(div_known(s.fast_env - s.slow_env, 7.0, 1.0 / 7.0), 0.0, 1.0);
    const double syllabic = dclamp(sustained * sustained * (1.0 - 0.35 * transient_bias), 0.0, 1.0);
    s.target_release_ms = 50.0 + syllabic * (400.0 - 50.0);
  } else {
    s.target_release_ms = p.base_release_ms;
  }
  const double release_diff = s.target_release_ms - s.cur_release_ms;
  if (fabs(release_diff) > 1.0) {
    s.cur_release_ms = p.release_smoothing_coeff * s.cur_release_ms +
                       (1.0 - p.release_smoothing_coeff) * s.target_release_ms;
  } else {
    s.cur_release_ms = s.target_release_ms;
  }

  const double target = comp_gain_reduction(p, detector_db);
  // smooth_gain_reduction, compressor.rs:468-505
  if (!p.adaptive_release) {
    const double k = target > s.gr ? p.attack_coeff : s.release_coeff;
    s.gr = k * s.gr + (1.0 - k) * target;
    s.fast_env = s.gr;
    s.slow_env = 0.0;
  } else {
    if (target > s.gr) {
      s.fast_env = p.attack_coeff * s.gr + (1.0 - p.attack_coeff) * target;
    } else {
      s.fast_env = p.fast_release_coeff * s.fast_env + (1.0 - p.fast_release_coeff) * target;
    }
    if (target > 3.0) {
      s.slow_env = p.slow_charge_coeff * s.slow_env + (1.0 - p.slow_charge_coeff) * target;
    } else {
      s.slow_env *= p.slow_release_coeff;
    }
    s.gr = fmax(s.fast_env, s.slow_env);
  }
  const double output_gain = db2lin(-s.gr) * makeup_lin;
  return (float)(x * output_gain);
}

// ------------------------------------------------------------------ true-peak FIR
// Bandlimited4xPeak::observe, dsp/true_peak.rs:173-186; h[k] = sample n-k lives at row (row0 - k)
__device__ __forceinline__ float tp_observe(const float (*buf)[kLanes], int row0, int lane) {
  float h[kTpTaps];
#pragma unroll
  for (int k = 0; k < kTpTaps; ++k) h[k] = buf[row0 - k][lane];
  float peak = fabsf(h[0]);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < kTpTaps; ++k) acc = __builtin_fmaf(AF_TP_FIR[p][k], h[k], acc);
    peak = fmaxf(peak, fabsf(acc));
  }
  return peak;
}

// =====================================================================================
// Kernel 1: lane per stream
// =====================================================================================
extern "C" __global__ __launch_bounds__(kLanes) void chain_lane_kernel(LaunchArgs a) {
  extern __shared__ float lim_lds[];  // [2*W][kLanes]: ring rows then suffix-max rows
  __shared__ TileLds L;

  const ChainParams &P = *a.params;
  const int lane = threadIdx.x;
  const int s0 = blockIdx.x * kLanes;
  const int s = s0 + lane;
  const bool valid = s < a.n_streams;
  const int sc = valid ? s : a.n_streams - 1;
  const int64_t NS = a.n_streams;
  const uint32_t flags = P.flags;
  const int nsec = (flags & kFlagEq) ? P.n_eq_sections : 0;
  const int W = P.lim.lookahead_samples + 1;
  float(*ring)[kLanes] = reinterpret_cast<float(*)[kLanes]>(lim_lds);
  float(*suf)[kLanes] = reinterpret_cast<float(*)[kLanes]>(lim_lds + (size_t)W * kLanes);

  // ---- restore the LDS-resident per-stream state
  for (int r = 0; r < kTpTaps; ++r) {
    L.tpi[r][lane] = a.st32[(int64_t)(kTpInHist + r) * NS + sc];
    L.tpo[r][lane] = a.st32[(int64_t)(kTpOutHist + r) * NS + sc];
  }
  if (flags & kFlagLimiter) {
    for (int r = 0; r < 2 * W; ++r) lim_lds[(size_t)r * kLanes + lane] = a.st32[(int64_t)(kLimRing + r) * NS + sc];
  }
  // ---- register-resident state
  float dc_x1 = a.st32[(int64_t)kDcX1 * NS + sc];
  float dc_y1 = a.st32[(int64_t)kDcY1 * NS + sc];
  float tp_gain = a.st32[(int64_t)kTpGain * NS + sc];
  float lim_prefix = a.st32[(int64_t)kLimPrefix * NS + sc];
  double pre_z1 = a.st64[(int64_t)kPreZ1 * NS + sc];
  double pre_z2 = a.st64[(int64_t)kPreZ2 * NS + sc];
  double lim_gain = a.st64[(int64_t)kLimGain * NS + sc];
  CompState cs;
  cs.sc_prev_in = a.st64[(int64_t)kCompScPrevIn * NS + sc];
  cs.sc_prev_out = a.st64[(int64_t)kCompScPrevOut * NS + sc];
  cs.low_env = a.st64[(int64_t)kCompLowEnv * NS + sc];
  cs.voiced_env = a.st64[(int64_t)kCompVoicedEnv * NS + sc];
  cs.presence_env = a.st64[(int64_t)kCompPresenceEnv * NS + sc];
  cs.plosive = a.st64[(int64_t)kCompPlosive * NS + sc];
  cs.peak_env_db = a.st64[(int64_t)kCompPeakEnvDb * NS + sc];
  cs.rms_env_sq = a.st64[(int64_t)kCompRmsEnvSq * NS + sc];
  cs.gr = a.st64[(int64_t)kCompGr * NS + sc];
  cs.fast_env = a.st64[(int64_t)kCompFastEnv * NS + sc];
  cs.slow_env = a.st64[(int64_t)kCompSlowEnv * NS + sc];
  cs.cur_release_ms = a.st64[(int64_t)kCompCurReleaseMs * NS + sc];
  cs.target_release_ms = a.st64[(int64_t)kCompTargetReleaseMs * NS + sc];
  cs.release_coeff = a.st64[(int64_t)kCompReleaseCoeff * NS + sc];
  cs.smoothed_makeup = a.st64[(int64_t)kCompSmoothedMakeup * NS + sc];
  __syncthreads();

  const int cb = P.control_block;
  int64_t done = 0;       // samples of this launch already processed
  int64_t block_index = 0;
  for (int64_t blk0 = 0; blk0 < a.n_samples; blk0 += cb, ++block_index) {
    const int blk_len = (int)((a.n_samples - blk0) < cb ? (a.n_samples - blk0) : cb);
    float in_peak = 0.0f, out_peak = 0.0f, tp_in_peak = 0.0f, out_tp = 0.0f, tp_gmin = 1.0f;
    double in_sq = 0.0, out_sq = 0.0, lim_gmin = 1.0;
    uint32_t tp_limited = 0, non_finite = 0;
    // smoothed makeup is constant inside a block (compressor.rs:771-772 vs 721)
    const double makeup_lin = db2lin(cs.smoothed_makeup);
    // TruePeakLimiter::set_ceiling_linear(10f32.powf(ceiling_db as f32 / 20)), block_processor.rs:150-151
    const float tp_ceiling = P.tp.ceiling_linear;

    for (int t0 = 0; t0 < blk_len; t0 += kTile) {
      const int len = (blk_len - t0) < kTile ? (blk_len - t0) : kTile;
      const int64_t abs0 = blk0 + t0;
      load_tile(a, L, abs0, len, s0, lane);
      __syncthreads();

      // ---- input scrub / clamp / block input stats (python_api.rs:515-523, routing.rs:802-823)
      for (int t = 0; t < len; ++t) {
        float v = L.x[t][lane];
        if ((flags & (kFlagInputScrub | kFlagInputClamp)) && !finite_f32(v)) v = 0.0f;
        if (flags & kFlagInputClamp) v = fclamp(v, -1.0f, 1.0f);
        L.x[t][lane] = v;
        in_sq += (double)v * (double)v;
        in_peak = fmaxf(in_peak, fabsf(v));
      }
      // ---- DC block + fixed high-pass (routing.rs:826-843)
      if (flags & kFlagDcBlock) {
        const BiquadCoef c = P.pre_hp;
        for (int t = 0; t < len; ++t) {
          const float in = L.x[t][lane];
          const float o = in - dc_x1 + 0.995f * dc_y1;
          dc_x1 = in;
          dc_y1 = o;
          float r = o;
          if (flags & kFlagPreHighpass) {
            const double xin = (double)o;
            const double y = c.b0 * xin + pre_z1;
            pre_z1 = c.b1 * xin - c.a1 * y + pre_z2;
            pre_z2 = c.b2 * xin - c.a2 * y;
            r = (float)y;
          }
          L.x[t][lane] = r;
        }
      }
      // ---- 10-band EQ, band-major over the tile (eq.rs:371-379)
      for (int k = 0; k < nsec; ++k) {
        const SectionParams &sp = P.eq[k];
        const int64_t base = (int64_t)(kEqBase + 4 * k) * NS + sc;
        double z1 = a.st64[base], z2 = a.st64[base + NS];
        double pz1 = 0.0, pz2 = 0.0;
        int rem = sp.xf_remaining - (int)(done < sp.xf_remaining ? done : sp.xf_remaining);
        if (rem > 0) {
          pz1 = a.st64[base + 2 * NS];
          pz2 = a.st64[base + 3 * NS];
        }
        biquad_tile(L.x, lane, len, sp, rem, z1, z2, pz1, pz2);
        if (valid) {
          a.st64[base] = z1;
          a.st64[base + NS] = z2;
          if (rem > 0) {
            a.st64[base + 2 * NS] = pz1;
            a.st64[base + 3 * NS] = pz2;
          }
        }
      }
      // ---- compressor (compressor.rs:700-722)
      if (flags & kFlagCompressor) {
        for (int t = 0; t < len; ++t) L.x[t][lane] = comp_sample(P.comp, cs, L.x[t][lane], makeup_lin);
      }
      // ---- lookahead limiter (limiter.rs:246-284) -> tpi rows
      if (flags & kFlagLimiter) {
        const double ceil_lin = P.lim.ceiling_linear;
        const double rc = P.lim.release_coeff;
        int j = (int)((a.samples_before + abs0) % W);
        for (int t = 0; t < len; ++t) {
          const float xin = L.x[t][lane];
          const float ax = fabsf(xin);
          const int jn = (j + 1 == W) ? 0 : j + 1;
          const float delayed = ring[jn][lane];
          const float sfx = (j + 1 < W) ? suf[j + 1][lane] : 0.0f;
          lim_prefix = (j == 0) ? ax : fmaxf(lim_prefix, ax);
          const double peak = (double)fmaxf(sfx, lim_prefix);
          ring[j][lane] = xin;
          if (j + 1 == W) {  // block of W inputs complete: suffix maxima for the next block
            float m = 0.0f;
            for (int k = W - 1; k >= 0; --k) {
              m = fmaxf(m, fabsf(ring[k][lane]));
              suf[k][lane] = m;
            }
          }
          j = jn;
          const double target = peak > ceil_lin ? ceil_lin / peak : 1.0;
          if (target < lim_gain) {
            lim_gain = target;
          } else {
            lim_gain = rc * lim_gain + (1.0 - rc) * target;
          }
          lim_gmin = fmin(lim_gmin, lim_gain);
          const double limited = (double)delayed * lim_gain;
          L.tpi[kTpTaps + t][lane] = (float)dclamp(limited, -ceil_lin, ceil_lin);
        }
        // ---- 4x true-peak limiter (true_peak.rs:337-378) -> tpo rows
        const float rel = P.tp.release_coeff;
        for (int t = 0; t < len; ++t) {
          float input = L.tpi[kTpTaps + t][lane];
          if (!finite_f32(input)) {
            input = 0.0f;
            L.tpi[kTpTaps + t][lane] = 0.0f;
          }
          const float delayed = L.tpi[kTpTaps + t - kTpDelay][lane];
          const float itp = tp_observe(L.tpi, kTpTaps + t, lane);
          tp_in_peak = fmaxf(tp_in_peak, itp);
          float target = 1.0f;
          if (itp > tp_ceiling) target = fclamp((tp_ceiling * 0.999f) / itp, 0.0f, 1.0f);
          if (target < tp_gain) {
            tp_gain = target;
            tp_limited = 1;
          } else {
            tp_gain = rel * tp_gain + (1.0f - rel) * target;
          }
          tp_gmin = fminf(tp_gmin, tp_gain);
          float o = fclamp(delayed * tp_gain, -tp_ceiling, tp_ceiling);
          if (!finite_f32(o)) o = 0.0f;
          L.tpo[kTpTaps + t][lane] = o;
        }
      } else {
        for (int t = 0; t < len; ++t) L.tpo[kTpTaps + t][lane] = L.x[t][lane];
      }
      // ---- block output stats + true-peak detector (block_processor.rs:158-159)
      for (int t = 0; t < len; ++t) {
        const float o = L.tpo[kTpTaps + t][lane];
        L.x[t][lane] = o;
        if (finite_f32(o)) {
          out_sq += (double)o * (double)o;
        } else {
          non_finite = 1;
          L.tpo[kTpTaps + t][lane] = 0.0f;  // TruePeakDetector feeds 0 for non-finite, true_peak.rs:212
        }
        out_peak = fmaxf(out_peak, fabsf(o));
        out_tp = fmaxf(out_tp, tp_observe(L.tpo, kTpTaps + t, lane));
      }
      __syncthreads();
      store_tile(a, L, abs0, len, s0, lane);
      // ---- slide the 32-sample histories
      for (int r = 0; r < kTpTaps; ++r) {
        const float vi = L.tpi[len + r][lane];
        const float vo = L.tpo[len + r][lane];
        L.tpi[r][lane] = vi;
        L.tpo[r][lane] = vo;
      }
      __syncthreads();
      done += len;
    }

    // ---- end of block: update_auto_makeup_gain with auto-makeup off (compressor.rs:604-617)
    if (flags & kFlagCompressor) {
      const double makeup_coeff = pow(P.comp.makeup_smoothing_coeff, (double)(blk_len < 1 ? 1 : blk_len));
      const double target = P.comp.makeup_gain_db;
      const double diff = target - cs.smoothed_makeup;
      if (fabs(diff) > 0.1) {
        cs.smoothed_makeup = makeup_coeff * cs.smoothed_makeup + (1.0 - makeup_coeff) * target;
      } else {
        cs.smoothed_makeup = target;
      }
    } else {
      cs.gr = 0.0;  // compressor.rs:705-708
    }
    if (valid && a.stats) {
      BlockStats st;
      st.input_sample_peak = in_peak;
      st.output_sample_peak = out_peak;
      st.tp_limiter_input_peak = tp_in_peak;
      st.output_true_peak = out_tp;
      // peak_gain_reduction_db = max over the block of -lin2db(g) (limiter.rs:273-280): the max sits at min g
      st.limiter_peak_gr_db = (flags & kFlagLimiter) && lim_gmin < 1.0 ? (float)(-lin2db(lim_gmin, 1e-10)) : 0.0f;
      // true_peak.rs:315-321, 363-365
      st.tp_limiter_gr_db =
          (flags & kFlagLimiter) && tp_gmin < 1.0f ? -20.0f * log10f(fmaxf(tp_gmin, 1e-10f)) : 0.0f;
      st.compressor_gr_db = (flags & kFlagCompressor) ? (float)cs.gr : 0.0f;
      st.deesser_gr_db = 0.0f;
      st.input_square_sum = in_sq;
      st.output_square_sum = out_sq;
      st.tp_limited_events = tp_limited;
      st.non_finite_output = non_finite;
      st.makeup_gain_db = (flags & kFlagCompressor) ? (float)cs.smoothed_makeup : 0.0f;
      st.makeup_activity = 0.0f;
      st.makeup_reliability = 0.0f;
      st.pad = 0.0f;
      a.stats[block_index * NS + s] = st;
    }
  }

  // ---- save state
  if (valid) {
    for (int r = 0; r < kTpTaps; ++r) {
      a.st32[(int64_t)(kTpInHist + r) * NS + s] = L.tpi[r][lane];
      a.st32[(int64_t)(kTpOutHist + r) * NS + s] = L.tpo[r][lane];
    }
    if (flags & kFlagLimiter) {
      for (int r = 0; r < 2 * W; ++r) a.st32[(int64_t)(kLimRing + r) * NS + s] = lim_lds[(size_t)r * kLanes + lane];
    }
    if (flags & kFlagDcBlock) {  // otherwise the front-end rows belong to whoever runs the front end
      a.st32[(int64_t)kDcX1 * NS + s] = dc_x1;
      a.st32[(int64_t)kDcY1 * NS + s] = dc_y1;
      a.st64[(int64_t)kPreZ1 * NS + s] = pre_z1;
      a.st64[(int64_t)kPreZ2 * NS + s] = pre_z2;
    }
    a.st32[(int64_t)kTpGain * NS + s] = tp_gain;
    a.st32[(int64_t)kLimPrefix * NS + s] = lim_prefix;
    a.st64[(int64_t)kLimGain * NS + s] = lim_gain;
    a.st64[(int64_t)kCompScPrevIn * NS + s] = cs.sc_prev_in;
    a.st64[(int64_t)kCompScPrevOut * NS + s] = cs.sc_prev_out;
    a.st64[(int64_t)kCompLowEnv * NS + s] = cs.low_env;
    a.st64[(int64_t)kCompVoicedEnv * NS + s] = cs.voiced_env;
    a.st64[(int64_t)kCompPresenceEnv * NS + s] = cs.presence_env;
    a.st64[(int64_t)kCompPlosive * NS + s] = cs.plosive;
    a.st64[(int64_t)kCompPeakEnvDb * NS + s] = cs.peak_env_db;
    a.st64[(int64_t)kCompRmsEnvSq * NS + s] = cs.rms_env_sq;
    a.st64[(int64_t)kCompGr * NS + s] = cs.gr;
    a.st64[(int64_t)kCompFastEnv * NS + s] = cs.fast_env;
    a.st64[(int64_t)kCompSlowEnv * NS + s] = cs.slow_env;
    a.st64[(int64_t)kCompCurReleaseMs * NS + s] = cs.cur_release_ms;
    a.st64[(int64_t)kCompTargetReleaseMs * NS + s] = cs.target_release_ms;
    // compressor.rs:760-761 (the per-sample recompute, materialised once)
    const double tau = fmax(cs.cur_release_ms, 0.001) / 1000.0;
    a.st64[(int64_t)kCompReleaseCoeff * NS + s] =
        P.comp.adaptive_release ? exp(-1.0 / (tau * P.comp.sample_rate)) : cs.release_coeff;
    a.st64[(int64_t)kCompSmoothedMakeup * NS + s] = cs.smoothed_makeup;
  }
}

// ------------------------------------------------------------------ launcher
size_t lane_kernel_dynamic_lds(int lookahead_samples) {
  return (size_t)2 * (lookahead_samples + 1) * kLanes * sizeof(float);
}

hipError_t launch_chain_lane(const LaunchArgs &args, int lookahead_samples, hipStream_t stream) {
  const int groups = (args.n_streams + kLanes - 1) / kLanes;
  const size_t dyn = lane_kernel_dynamic_lds(lookahead_samples);
  hipLaunchKernelGGL(chain_lane_kernel, dim3(groups), dim3(kLanes), dyn, stream, args);
  return hipGetLastError();
}

}  // namespace af
