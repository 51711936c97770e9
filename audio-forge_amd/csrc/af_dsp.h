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
__device__ __forceinline__ double db2lin(double db) { return exp10(db / 20.0); }

__device__ __forceinline__ bool finite_f32(float v) {
  return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u;
}

// Compressor::compute_gain_reduction, dsp/compressor.rs:657-678
__device__ __forceinline__ double comp_gain_reduction(const CompressorParams &p, double detector_db) {
  const double comp_factor = 1.0 - 1.0 / p.ratio;
  if (p.knee_db <= 0.0) {
    if (detector_db <= p.threshold_db) return 0.0;
    return (detector_db - p.threshold_db) * comp_factor;
  }
  const double knee_half = p.knee_db / 2.0;
  const double knee_start = p.threshold_db - knee_half;
  const double knee_end = p.threshold_db + knee_half;
  if (detector_db <= knee_start) return 0.0;
  if (detector_db >= knee_end) return (detector_db - p.threshold_db) * comp_factor;
  const double x = detector_db - knee_start;
  return comp_factor * x * x / (2.0 * p.knee_db);
}

}  // namespace af
