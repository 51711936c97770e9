// af_deesser_math.h -- the de-esser's per-sample arithmetic (rust-core/src/dsp/deesser.rs), shared by the lane-per-stream pass
// (af_deesser.hip) and the stage pipeline's de-esser stages (af_stages.hip): the same expressions, operation for operation.
#pragma once
#include <hip/hip_runtime.h>

#include "af_dsp.h"

namespace af {
namespace deess {

constexpr double kVoiceRefDiscount = 0.6;  // deesser.rs:19-31
constexpr double kRatioGateDb = 1.5, kRatioFullDb = 10.0;
constexpr double kLevelGateDb = -62.0, kLevelFullDb = -24.0;
constexpr double kVoiceGateDb = -58.0, kVoiceFullDb = -34.0;
constexpr double kNarrowGate = 0.34, kNarrowFull = 0.68;

struct Bq {
  double z1, z2, pz1, pz2;
};

struct BandState {
  double env, confidence, baseline, reduction, gain_db, cancelled;
  BiquadCoef dyn;
  Bq hp, lp, eq;
};

__device__ __forceinline__ double direct(const BiquadCoef &c, double x, double &z1, double &z2) {  // biquad.rs:263-274
  const double y = c.b0 * x + z1;
  z1 = c.b1 * x - c.a1 * y + z2;
  z2 = c.b2 * x - c.a2 * y;
  return y;
}

// Biquad::process_sample (biquad.rs:290-327) for a filter whose crossfade is stream-uniform
__device__ __forceinline__ float section_sample(const SectionParams &sp, int rem, float xin, Bq &s) {
  const double x = (double)xin;
  if (rem > 0) {
    const double ya = direct(sp.active, x, s.z1, s.z2);
    const double yp = direct(sp.pending, x, s.pz1, s.pz2);
    const double fade = (double)(sp.xf_total - rem + 1) / (double)sp.xf_total;
    const double y = ya * (1.0 - fade) + yp * fade;
    if (rem == 1) {  // promote_pending_coefficients, biquad.rs:276-286
      s.z1 = s.pz1;
      s.z2 = s.pz2;
    }
    return (float)y;
  }
  return (float)direct(sp.xf_remaining > 0 ? sp.pending : sp.active, x, s.z1, s.z2);
}

__device__ __forceinline__ double smooth_value(double prev, double input, double attack, double release) {  // :149-157
  const double c = input > prev ? attack : release;
  return c * prev + (1.0 - c) * input;
}
__device__ __forceinline__ double lerp(double a, double b, double t) { return a + (b - a) * t; }
__device__ __forceinline__ double normalize_range(double v, double start, double end) {
  return dclamp((v - start) / (end - start), 0.0, 1.0);
}

// deesser.rs:173-224
__device__ __forceinline__ double confidence_target(double side_db, double voice_db, double narrowness) {
  const double ratio_db = fmax(side_db - voice_db, 0.0);
  const double ratio_conf = normalize_range(ratio_db, kRatioGateDb, kRatioFullDb);
  const double level_conf = normalize_range(side_db, kLevelGateDb, kLevelFullDb);
  const double voice_conf = normalize_range(voice_db, kVoiceGateDb, kVoiceFullDb);
  const double narrow_support = (ratio_db > 6.0 && side_db > -45.0) ? 0.75 : 0.0;
  const double voice_support = fmax(voice_conf, narrow_support);
  const double balance = ratio_conf > 0.12 ? fmax(ratio_conf, voice_support * 0.65) : ratio_conf;
  const double penalty = lerp(0.35, 1.0, balance);
  const double narrow_gain = lerp(0.35, 1.0, normalize_range(narrowness, kNarrowGate, kNarrowFull));
  return (0.62 * ratio_conf + 0.18 * level_conf + 0.20 * voice_support) * penalty * narrow_gain;
}

// Biquad::calculate_coefficients for Peaking (biquad.rs:109-182) at a fixed centre / Q
__device__ __forceinline__ BiquadCoef peaking(double cos_omega, double alpha, double gain_db) {
  const double a = exp10(gain_db / 40.0);
  const double b0 = 1.0 + alpha * a, b1 = -2.0 * cos_omega, b2 = 1.0 - alpha * a;
  const double a0 = 1.0 + alpha / a, a2 = 1.0 - alpha / a;
  return BiquadCoef{b0 / a0, b1 / a0, b2 / a0, b1 / a0, a2 / a0};
}


}  // namespace deess
}  // namespace af
