// af_device.h -- plain structs shared by the host configuration mirror and the HIP kernels.
//
// Everything a stream needs that is identical for all streams of an engine (coefficients,
// time constants, switches) lives in ChainParams and is read through scalar loads; what
// differs per stream (filter memories, envelopes, delay lines) lives in two
// structure-of-arrays state planes, f64 and f32, laid out [field][stream] so that lane i
// of a wavefront touches consecutive addresses.
#pragma once
#include <stdint.h>

namespace af {

constexpr int kNumBands = 10;
constexpr int kMaxSectionsPerBand = 4;
constexpr int kMaxEqSections = kNumBands * kMaxSectionsPerBand;
constexpr int kTpTaps = 32;
constexpr int kTpDelay = 20;
constexpr int kMaxLookahead = 1024;
constexpr int kTile = 64;          // samples per LDS tile (per stream)
constexpr int kLanes = 64;         // streams per workgroup in the lane-per-stream kernel
constexpr int kLdsLookaheadMax = 127;  // limiter ring kept in LDS up to this lookahead

struct BiquadCoef {
  double b0, b1, b2, a1, a2;
};

// One second-order section with the reference's parallel-state coefficient crossfade
// (dsp/biquad.rs:39-66).  xf_* are identical for all streams because streams advance in
// lock step.
struct SectionParams {
  BiquadCoef active;
  BiquadCoef pending;
  int32_t xf_total;
  int32_t xf_remaining;  // at the start of the launch
};

// dsp/compressor.rs:46-129, the parameter half
struct CompressorParams {
  double threshold_db, ratio, knee_db;
  double attack_coeff, detector_release_coeff, rms_coeff;
  double release_smoothing_coeff, base_release_ms;
  double band_env_coeff;        // tc(SIDECHAIN_BAND_ENV_MS), compressor.rs:429
  double fast_release_coeff;    // tc(50 ms),  compressor.rs:482-483
  double slow_charge_coeff;     // tc(250 ms), compressor.rs:484-485
  double slow_release_coeff;    // tc(400 ms), compressor.rs:486-487
  double sidechain_highpass_coeff;
  double makeup_gain_db, makeup_smoothing_coeff, makeup_silence_relax_coeff;
  double speech_activity_smoothing_coeff, target_lufs, noise_reference_reliability;
  double sample_rate;
  // host-evaluated pieces of compute_gain_reduction (compressor.rs:657-678); same IEEE ops, done once
  double comp_factor, knee_start, knee_end, two_knee, two_knee_recip;
  // auto-makeup (compressor.rs:598-653) and its momentary-loudness meter (loudness.rs:99-135)
  double kw_b[5], kw_a[5];          // K-weighting as one 4th-order section (BS.1770-4)
  double makeup_pow_cb, relax_pow_cb, activity_pow_cb;  // coeff^control_block, host pow()
  double meter_frames;              // samples in the 400 ms window
  double vad_reliability, noise_floor_db, live_noise_reliability;  // AutoMakeupActivityInput, compressor.rs:32-37
  int32_t meter_slots;              // control blocks per 400 ms window (0 = meter unavailable)
  int32_t has_evidence;
  int32_t adaptive_release, sidechain_highpass_enabled, auto_makeup_enabled, pad;
};

// dsp/limiter.rs:71-97
struct LimiterParams {
  double ceiling_db, ceiling_linear, release_coeff;
  int32_t lookahead_samples, pad;
};

// dsp/true_peak.rs:250-264
struct TruePeakParams {
  float ceiling_linear, release_coeff;
};

// dsp/deesser.rs:34-107: three sibilance bands, each a detector (HP -> LP) and a dynamic peaking EQ.
// The detector filters and the dynamic EQ are `Biquad`s and carry the usual coefficient crossfade
// (set_low_cut_hz / set_high_cut_hz schedule one); the dynamic EQ recomputes its RBJ peaking coefficients
// whenever the smoothed reduction moved by more than 0.001 dB (set_gain_db_immediate, deesser.rs:536-538).
struct DeEsserBandParams {
  SectionParams detector_hp, detector_lp, dynamic_eq;
  double dyn_cos_omega, dyn_alpha;  // target centre / Q of the dynamic EQ (deesser.rs:263-272)
};
struct DeEsserParams {
  double attack_coeff, release_coeff, detector_attack_coeff, detector_release_coeff;
  double max_reduction_db, threshold_db, ratio, auto_amount;
  double baseline_fall, baseline_rise, baseline_inactive;  // tc(13.88 / 34.72 / 20.82 ms), deesser.rs:274-287
  int32_t auto_enabled, pad;
  DeEsserBandParams bands[3];
};

enum ChainFlags : uint32_t {
  kFlagDeesser = 1u << 0,
  kFlagEq = 1u << 1,
  kFlagCompressor = 1u << 2,
  kFlagLimiter = 1u << 3,
  kFlagEqBeforeDeesser = 1u << 4,
  kFlagInputScrub = 1u << 5,   // python_api.rs:517-520
  kFlagInputClamp = 1u << 6,   // routing.rs:802-823
  kFlagDcBlock = 1u << 7,      // routing.rs:826-843
  kFlagPreHighpass = 1u << 8,
  kFlagPrePass = 1u << 9,      // first of two launches: front end + EQ only, no detector, no compressor bookkeeping
  kFlagInputDone = 1u << 10,   // the input unit's work (block input statistics) was done by another kernel (af_eq_systolic.hip)
  kFlagCompOnly = 1u << 11,    // the launch ends at the compressor's output: limiter, true-peak limiter, output statistics and
                               // detector belong to another kernel (af_roles.hip), which owns their state rows
};

struct ChainParams {
  uint32_t flags;
  int32_t n_eq_sections;
  int32_t control_block;
  int32_t pad;
  BiquadCoef pre_hp;  // Biquad(HighPass, 80 Hz, Q 0.707), processor.rs:74-76
  SectionParams eq[kMaxEqSections];
  CompressorParams comp;
  LimiterParams lim;
  TruePeakParams tp;
  DeEsserParams deesser;
};

// ---- per-stream state planes -------------------------------------------------------
// f64 plane field indices
enum F64Field : int {
  kPreZ1 = 0, kPreZ2,
  kCompScPrevIn, kCompScPrevOut, kCompLowEnv, kCompVoicedEnv, kCompPresenceEnv, kCompPlosive,
  kCompPeakEnvDb, kCompRmsEnvSq, kCompGr, kCompFastEnv, kCompSlowEnv,
  kCompCurReleaseMs, kCompTargetReleaseMs, kCompReleaseCoeff, kCompSmoothedMakeup,
  kCompActivityScore, kCompActivityReliability, kCompCurrentLufs,
  kLimGain,
  kDeBroadbandEnv, kDeCurrentReduction, kDeConfidence,
  // per de-esser band (x3): env, confidence, baseline, reduction, dyn gain_db, dyn crossfade-cancelled flag,
  // the dynamic EQ's five live coefficients, then z1 z2 pz1 pz2 for detector_hp, detector_lp, dynamic_eq
  kDeBand0,
  kDeBandStride = 23,
  kEqBase = kDeBand0 + 3 * kDeBandStride,  // then 4 per section: z1 z2 pz1 pz2
  kF64Fixed = kEqBase
};
// after the EQ sections: the loudness meter (only when auto-makeup is on)
enum MeterField : int { kMeterV1 = 0, kMeterV2, kMeterV3, kMeterV4, kMeterPos, kMeterRing, kMeterFixed = kMeterRing };
inline int f64_field_count(int n_sections, int meter_slots = 0) {
  return kF64Fixed + 4 * n_sections + (meter_slots > 0 ? kMeterFixed + meter_slots : 0);
}

// f32 plane field indices
enum F32Field : int {
  kDcX1 = 0, kDcY1,
  kTpGain,
  kLimPrefix,            // running prefix max of the current van-Herk block
  kTpInHist,             // 32 rows: last 32 true-peak-limiter inputs (oldest first)
  kTpOutHist = kTpInHist + kTpTaps,  // 32 rows: last 32 outputs
  kLimRing = kTpOutHist + kTpTaps,   // W rows ring + W rows suffix-max, W = lookahead+1
  kF32Fixed = kLimRing
};
inline int f32_field_count(int lookahead) { return kF32Fixed + 2 * (lookahead + 1); }

// ---- per-block statistics row (== af_block_stats in include/audioforge_mi.h) --------
struct BlockStats {
  float input_sample_peak, output_sample_peak, tp_limiter_input_peak, output_true_peak;
  float limiter_peak_gr_db, tp_limiter_gr_db, compressor_gr_db, deesser_gr_db;
  double input_square_sum, output_square_sum;
  uint32_t tp_limited_events, non_finite_output;
  float makeup_gain_db, makeup_activity, makeup_reliability, pad;  // compressor metering at block end
};

struct LaunchArgs {
  const ChainParams *params;  // device; an array when `group_preset` is set
  const int32_t *group_preset;  // device, [ceil(n_streams / 64)]: the parameter block of each 64-stream group, or null (all use [0])
  double *st64;               // [f64 fields][n_streams]
  float *st32;                // [f32 fields][n_streams]
  const float *in;
  float *out;
  BlockStats *stats;          // [blocks][n_streams]
  int32_t *status;            // device word: non-zero when a kernel gave up on a token (never expected)
  const BlockStats *pre_stats;  // rows of the pre-pass launch (compressor-input block power), or null
  const double *pre_power;    // [blocks][n_streams] the same block power as a plain array (af_eq_systolic.hip as the pre-pass); wins over pre_stats
  const double *vad_prob;     // [blocks][n_streams] speech posteriors for auto-makeup, or null
  const int64_t *ready;       // device counter or null: samples of `in` (launch-relative, every stream) that producers on other
                              // streams have finished so far.  Set: the launch covers a whole call and follows the suppressor's
                              // windows as they arrive (a chunk waits until the counter covers it) -- ONE launch per call
  int64_t n_samples;
  int64_t stream_stride;
  int64_t samples_before;     // samples processed by earlier launches (van-Herk phase)
  int32_t n_streams;
  int32_t layout;
};

}  // namespace af
