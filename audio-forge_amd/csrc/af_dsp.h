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
// dsp/util.rs:18-20
__device__ __forceinline__ double lin2db(double linear, double floor_) {
  return 20.0 * log10(fmax(fabs(linear), floor_));
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
