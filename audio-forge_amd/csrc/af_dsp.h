// af_dsp.h -- device helpers shared by the chain kernels (arithmetic restated from
// rust-core/src/dsp/*.rs; the citing comments sit on each function).
#pragma once
#include <hip/hip_runtime.h>

#include "af_device.h"
#include "tp_fir_table.h"

namespace af {

// ------------------------------------------------------------------ small helpers
__device__ __forceinline__ double dclamp(double x, double lo, double hi) {
  return x < lo ? lo : (x > hi ? hi : x);
}
__device__ __forceinline__ float fclamp(float x, float lo, float hi) {
  return x < lo ? lo : (x > hi ? hi : x);
}
// log10 for positive, finite, normal arguments (every caller clamps to a floor >= 1e-10 first).
// The device library's log10 is ~105 instructions (it also serves denormals, zero, negatives and keeps
// error < 1 ulp through double-double arithmetic); the compressor evaluates four per sample, which made it
// 40 % of the whole chain.  This is the classic fdlibm reduction (x = 2^k * m, m in [sqrt(1/2), sqrt(2)),
// s = (m-1)/(m+1), degree-14 odd minimax series in s) in ~40 instructions with error < 2 ulp -- the same
// distance two host libms are apart.
__device__ __forceinline__ double fast_log10_pos(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int k = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;
  k = low ? k - 1 : k;
  const double f = m - 1.0;
  const double den = 2.0 + f;
  double r = __builtin_amdgcn_rcp(den);
  r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
  double sq = f * r;
  sq = __builtin_fma(__builtin_fma(-den, sq, f), r, sq);  // s = f / (2 + f), correctly rounded
  const double z = sq * sq;
  const double w = z * z;
  const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                                        2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  // ln(x) = k ln2 + f - (hfsq - s (hfsq + R)), ln2 split so that k * ln2_hi is exact
  const double ln = __builtin_fma(dk, 6.93147180369123816490e-01,
                                  f - (hfsq - __builtin_fma(sq, hfsq + R, dk * 1.90821492927058770002e-10)));
  return ln * 4.34294481903251816668e-01;
}
// dsp/util.rs:18-20
__device__ __forceinline__ double lin2db(double linear, double floor_) {
  return 20.0 * fast_log10_pos(fmax(fabs(linear), floor_));
}
// dsp/util.rs:12-14
// The reference evaluates 10^(dB/20) with libm pow(10, y); exp10(y) is the same function with
// a much shorter device routine (both are accurate to <1 ulp, which is also how far two host
// libms differ from each other).
//
// x / b for a divisor known up front: q = x*(1/b); r = fma(-b, q, x); q' = fma(r, 1/b, q).
// With 1/b correctly rounded this is the correctly rounded quotient (Markstein), i.e. exactly
// the value `x / b` has in the reference, at 3 instructions instead of the ~14 of a general
// f64 division (verified exhaustively-at-random on the host for the divisors used here).
__device__ __forceinline__ double div_known(double x, double b, double recip_b) {
  const double q = x * recip_b;
  const double r = __builtin_fma(-b, q, x);
  return __builtin_fma(r, recip_b, q);
}
__device__ __forceinline__ double db2lin(double db) { return exp10(div_known(db, 20.0, 0.05)); }

__device__ __forceinline__ bool finite_f32(float v) {
  return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u;
}

// The detector's RMS level and side-chain weight between the two places the reference takes them to dB and back
// (update_sidechain_band_metrics / blended_detector_db, compressor.rs:438-449,681-686,744-750).  The reference computes
//     rms_db = lin2db(rms);  weight_db = lin2db(w);  blended = 0.6 db2lin(peak_db) + 0.4 db2lin(rms_db);
//     detector_db = lin2db(blended) + weight_db
// i.e. exp10(log10(rms)) and log10(blended) + log10(w).  Here the RMS level and the weight stay LINEAR:
//     blended = 0.6 db2lin(peak_db) + 0.4 max(rms, 1e-10);  detector_db = lin2db(max(blended, 1e-10) * w)
// -- the same real-valued function (w lies in [0.35, 1.15], above every floor), two log10 and one exp10 fewer per sample:
// ~105 of the chain's ~420 f64 instructions per sample, on a kernel whose SIMDs are saturated with f64 issue.
// The results differ from the literal form by the rounding of those three library calls (<= 2 ulp of f64 on detector_db),
// the same distance as between the reference's libm and this one; the parity tolerances are unchanged (deviation 7).
// -DAF_LITERAL_DETECTOR builds the literal form (same-box A/B).
__device__ __forceinline__ double detector_rms_level(double rms_env_sq) {
#ifdef AF_LITERAL_DETECTOR
  return lin2db(sqrt(rms_env_sq), 1e-10);  // dB
#else
  return fmax(sqrt(rms_env_sq), 1e-10);    // linear
#endif
}
__device__ __forceinline__ double detector_weight(double clamped_weight) {  // clamped_weight in [0.35, 1.15]; 1.0 without the side chain
#ifdef AF_LITERAL_DETECTOR
  return lin2db(clamped_weight, 1e-10);  // dB
#else
  return clamped_weight;                 // linear
#endif
}
constexpr double kDetectorUnitWeight =
#ifdef AF_LITERAL_DETECTOR
    0.0;
#else
    1.0;
#endif
__device__ __forceinline__ double detector_db(double peak_env_db, double rms_level, double weight) {
#ifdef AF_LITERAL_DETECTOR
  const double blended = 0.6 * db2lin(peak_env_db) + 0.4 * db2lin(rms_level);
  return lin2db(blended, 1e-10) + weight;
#else
  const double blended = 0.6 * db2lin(peak_env_db) + 0.4 * rms_level;
  return 20.0 * fast_log10_pos(fmax(fabs(blended), 1e-10) * weight);
#endif
}

// Compressor::compute_gain_reduction, dsp/compressor.rs:657-678
__device__ __forceinline__ double comp_gain_reduction(const CompressorParams &p, double detector_db) {
  if (p.knee_db <= 0.0) {
    if (detector_db <= p.threshold_db) return 0.0;
    return (detector_db - p.threshold_db) * p.comp_factor;
  }
  if (detector_db <= p.knee_start) return 0.0;
  if (detector_db >= p.knee_end) return (detector_db - p.threshold_db) * p.comp_factor;
  const double x = detector_db - p.knee_start;
  return div_known(p.comp_factor * x * x, p.two_knee, p.two_knee_recip);
}

}  // namespace af
