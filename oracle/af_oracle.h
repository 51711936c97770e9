/*
 * af_oracle.h -- CPU restatement of AudioForge's rust-core per-frame voice chain.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and there only as the checker / the timed CPU baseline.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference checkout, rust-core/src/...).  Arithmetic types, operation order and
 * f32/f64 rounding points follow the Rust text; build with -ffp-contract=off.
 *
 * Parity status:
 *   KAT-pinned    : biquad, EQ, de-esser, compressor, limiter, true-peak limiter /
 *                   detector, offline block processor (tests.rs:1784-1885 and the
 *                   tracked evaluation reports).
 *   measurement-pinned : product resampler (rubato 0.14.1 is not vendored; af_resampler.c
 *                   reproduces the reference's published 16-digit measurements of it).
 *   spec-restated : K-weighted momentary / integrated loudness (ebur128 0.1.10 is not
 *                   vendored), RNNoise core (nnnoiseless 0.5.2, af_rnnoise.c), noise gate
 *                   expander path (behavioural pins only) -- parity unpinned for those.
 */
#ifndef AF_ORACLE_H
#define AF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- biquad */
/* dsp/biquad.rs:22-31 */
typedef enum {
  AFO_BQ_LOWSHELF = 0,
  AFO_BQ_HIGHSHELF = 1,
  AFO_BQ_PEAKING = 2,
  AFO_BQ_NOTCH = 3,
  AFO_BQ_HIGHPASS = 4,
  AFO_BQ_LOWPASS = 5,
  AFO_BQ_BYPASS = 6
} afo_biquad_type;

/* dsp/biquad.rs:39-66 */
typedef struct {
  double b0, b1, b2, a1, a2;
  double pb0, pb1, pb2, pa1, pa2;
  double pz1, pz2;
  size_t xf_total, xf_remaining;
  double z1, z2;
  afo_biquad_type type;
  double frequency, gain_db, q, sample_rate;
  int enabled;
} afo_biquad;

void afo_biquad_init(afo_biquad *f, afo_biquad_type type, double frequency, double gain_db,
                     double q, double sample_rate);
void afo_biquad_coefficients(const afo_biquad *f, double out[5]);
float afo_biquad_process_sample(afo_biquad *f, float input);
void afo_biquad_process_block(afo_biquad *f, float *buf, size_t n);
void afo_biquad_reset(afo_biquad *f);
void afo_biquad_set_frequency(afo_biquad *f, double frequency);
void afo_biquad_set_gain_db(afo_biquad *f, double gain_db);
void afo_biquad_set_gain_db_immediate(afo_biquad *f, double gain_db);
void afo_biquad_set_q(afo_biquad *f, double q);
void afo_biquad_set_parameters(afo_biquad *f, afo_biquad_type type, double frequency,
                               double gain_db, double q);
void afo_biquad_set_parameters_immediate(afo_biquad *f, afo_biquad_type type, double frequency,
                                         double gain_db, double q);
double afo_biquad_target_magnitude_db(const afo_biquad *f, double frequency_hz);

/* -------------------------------------------------------------------- EQ */
#define AFO_NUM_BANDS 10
#define AFO_MAX_PASS_SECTIONS 4

/* dsp/eq.rs:44-53 (stable public ids) */
typedef enum {
  AFO_EQ_LOW_SHELF = 0,
  AFO_EQ_BELL = 1,
  AFO_EQ_HIGH_SHELF = 2,
  AFO_EQ_NOTCH = 3,
  AFO_EQ_HIGH_PASS = 4,
  AFO_EQ_LOW_PASS = 5
} afo_eq_filter_type;

/* dsp/eq.rs:112-120 */
typedef struct {
  int32_t filter_type;
  double frequency_hz;
  double gain_db;
  double q;
  int32_t slope_db_per_octave;
  int32_t enabled;
} afo_eq_band_config;

typedef struct {
  afo_biquad sections[AFO_MAX_PASS_SECTIONS];
  afo_eq_band_config config;
  size_t processing_sections, target_sections;
} afo_eq_band;

typedef struct {
  afo_eq_band bands[AFO_NUM_BANDS];
  int enabled;
  double sample_rate;
} afo_eq;

void afo_eq_init(afo_eq *eq, double sample_rate);
void afo_eq_process_block(afo_eq *eq, float *buf, size_t n);
void afo_eq_reset(afo_eq *eq);
void afo_eq_set_band_gain(afo_eq *eq, size_t band, double gain_db);
void afo_eq_set_band_frequency(afo_eq *eq, size_t band, double frequency);
void afo_eq_set_band_q(afo_eq *eq, size_t band, double q);
void afo_eq_set_band_config(afo_eq *eq, size_t band, const afo_eq_band_config *config);
void afo_eq_magnitude_response_db(const afo_eq *eq, const double *freqs, size_t n, double *out);
/* returns 0 when valid, else writes a message (eq.rs:140-201) */
int afo_eq_band_config_validate(const afo_eq_band_config *c, size_t index, double sample_rate,
                                char *msg, size_t msg_len);

/* -------------------------------------------------------------- loudness */
/* dsp/loudness.rs:89-158 over ebur128 0.1.10 (Mode::M).  spec-restated. */
typedef struct {
  double b[5], a[5];
  double v[5];
  double *ring;          /* 400 ms of K-weighted samples */
  size_t ring_frames, ring_index;
  float current_lufs;
  uint32_t sample_rate;
  int valid;
} afo_loudness;

int afo_loudness_init(afo_loudness *m, uint32_t sample_rate);
void afo_loudness_free(afo_loudness *m);
void afo_loudness_process(afo_loudness *m, const float *samples, size_t n);
void afo_loudness_reset(afo_loudness *m);

/* ------------------------------------------------------------ compressor */
/* dsp/compressor.rs:32-37 */
typedef struct {
  double vad_probability, vad_reliability, noise_floor_db, live_noise_reliability;
} afo_auto_makeup_input;

/* dsp/compressor.rs:46-129 */
typedef struct {
  double threshold_db, ratio, attack_coeff, release_coeff, detector_release_coeff;
  double makeup_gain_db, makeup_gain_linear, knee_db;
  double peak_envelope_db, rms_envelope_sq, rms_coeff, current_gain_reduction_db;
  double sample_rate;
  int enabled, adaptive_release;
  double base_release_ms, current_release_ms, target_release_ms, release_smoothing_coeff;
  double fast_release_env_db, slow_release_env_db;
  afo_loudness meter;
  int has_meter;
  int auto_makeup_enabled;
  double target_lufs, smoothed_makeup_gain, makeup_smoothing_coeff, current_lufs;
  double speech_activity_score, speech_activity_smoothing_coeff;
  double auto_makeup_activity_reliability, noise_reference_reliability;
  double makeup_silence_relax_coeff;
  int sidechain_highpass_enabled;
  double sidechain_highpass_coeff, sidechain_highpass_prev_input, sidechain_highpass_prev_output;
  double low_band_env_sq, voiced_band_env_sq, presence_band_env_sq, plosive_ratio;
  double limiter_feedback_gain_reduction_db;
} afo_compressor;

void afo_compressor_init(afo_compressor *c, double threshold_db, double ratio, double attack_ms,
                         double release_ms, double makeup_gain_db, double knee_db,
                         double sample_rate);
void afo_compressor_free(afo_compressor *c);
void afo_compressor_set_threshold(afo_compressor *c, double v);
void afo_compressor_set_ratio(afo_compressor *c, double v);
void afo_compressor_set_attack_time(afo_compressor *c, double ms);
void afo_compressor_set_release_time(afo_compressor *c, double ms);
void afo_compressor_set_adaptive_release(afo_compressor *c, int enabled);
void afo_compressor_set_base_release_time(afo_compressor *c, double ms);
void afo_compressor_set_makeup_gain(afo_compressor *c, double db);
void afo_compressor_set_enabled(afo_compressor *c, int enabled);
void afo_compressor_set_auto_makeup_enabled(afo_compressor *c, int enabled);
void afo_compressor_set_target_lufs(afo_compressor *c, double v);
void afo_compressor_set_noise_reference_reliability(afo_compressor *c, double v);
void afo_compressor_set_sidechain_highpass_enabled(afo_compressor *c, int enabled);
void afo_compressor_set_limiter_feedback_gain_reduction_db(afo_compressor *c, double v);
float afo_compressor_process_sample(afo_compressor *c, float input);
void afo_compressor_process_block(afo_compressor *c, float *buf, size_t n,
                                  const afo_auto_makeup_input *evidence /* may be NULL */);
void afo_compressor_reset(afo_compressor *c);
double afo_compressor_compute_gain_reduction(const afo_compressor *c, double detector_db);
double afo_compressor_blended_detector_db(double peak_db, double rms_db);

/* --------------------------------------------------------------- limiter */
#define AFO_MAX_LOOKAHEAD 1024
/* dsp/limiter.rs:9-97 */
typedef struct {
  double ceiling_db, ceiling_linear, release_coeff, gain_reduction, peak_gain_reduction_db;
  double sample_rate;
  size_t lookahead_samples;
  float delay[AFO_MAX_LOOKAHEAD];
  uint64_t q_index[AFO_MAX_LOOKAHEAD];
  double q_value[AFO_MAX_LOOKAHEAD];
  size_t q_head, q_len;
  uint64_t next_input_index;
  size_t write_idx;
  int enabled;
} afo_limiter;

void afo_limiter_init(afo_limiter *l, double ceiling_db, double release_ms, double sample_rate,
                      double lookahead_ms);
void afo_limiter_set_ceiling(afo_limiter *l, double ceiling_db);
void afo_limiter_set_release_time(afo_limiter *l, double ms);
void afo_limiter_set_lookahead_ms(afo_limiter *l, double ms);
void afo_limiter_set_enabled(afo_limiter *l, int enabled);
float afo_limiter_process_sample(afo_limiter *l, float input);
void afo_limiter_process_block(afo_limiter *l, float *buf, size_t n);
double afo_limiter_peak_gain_reduction_and_reset(afo_limiter *l);
void afo_limiter_reset(afo_limiter *l);

/* ------------------------------------------------------------- true peak */
#define AFO_TP_TAPS 32
#define AFO_TP_LOOKAHEAD 20
/* dsp/true_peak.rs:156-186 */
typedef struct { float history[AFO_TP_TAPS]; } afo_tp_oversampler;
float afo_tp_observe(afo_tp_oversampler *o, float sample);

/* dsp/true_peak.rs:188-226 */
typedef struct { afo_tp_oversampler os; float last_peak; } afo_tp_detector;
void afo_tp_detector_init(afo_tp_detector *d);
float afo_tp_detector_process_block(afo_tp_detector *d, const float *samples, size_t n);

/* dsp/true_peak.rs:228-264 */
typedef struct {
  uint64_t limited_events;
  float input_true_peak, output_true_peak, max_gain_reduction_db;
} afo_tp_block_stats;

typedef struct {
  float ceiling_linear, release_coeff, gain_reduction;
  float delay[AFO_TP_LOOKAHEAD];
  size_t write_idx;
  afo_tp_oversampler in_os, out_os;
  float last_input_true_peak, last_output_true_peak, peak_gain_reduction_db, sample_rate;
} afo_tp_limiter;

void afo_tp_limiter_init(afo_tp_limiter *l, float sample_rate, float ceiling_db, float release_ms);
void afo_tp_limiter_set_ceiling_linear(afo_tp_limiter *l, float ceiling_linear);
void afo_tp_limiter_set_release_ms(afo_tp_limiter *l, float release_ms);
afo_tp_block_stats afo_tp_limiter_process_block(afo_tp_limiter *l, float *samples, size_t n);
void afo_tp_limiter_reset(afo_tp_limiter *l);

/* -------------------------------------------------------------- de-esser */
/* dsp/deesser.rs:34-107 */
typedef struct {
  double low_hz, high_hz, env, confidence, baseline_excess_db, reduction_db;
  afo_biquad detector_hp, detector_lp, dynamic_eq;
} afo_deesser_band;

typedef struct {
  int enabled, auto_enabled;
  double auto_amount, threshold_db, ratio, attack_coeff, release_coeff;
  double detector_attack_coeff, detector_release_coeff, max_reduction_db;
  double current_reduction_db, broadband_env, detector_confidence;
  double low_cut_hz, high_cut_hz, sample_rate;
  afo_deesser_band bands[3];
} afo_deesser;

void afo_deesser_init(afo_deesser *d, double sample_rate);
void afo_deesser_set_enabled(afo_deesser *d, int enabled);
void afo_deesser_set_auto_enabled(afo_deesser *d, int enabled);
void afo_deesser_set_auto_amount(afo_deesser *d, double v);
void afo_deesser_set_low_cut_hz(afo_deesser *d, double v);
void afo_deesser_set_high_cut_hz(afo_deesser *d, double v);
void afo_deesser_set_threshold_db(afo_deesser *d, double v);
void afo_deesser_set_ratio(afo_deesser *d, double v);
void afo_deesser_set_attack_ms(afo_deesser *d, double v);
void afo_deesser_set_release_ms(afo_deesser *d, double v);
void afo_deesser_set_max_reduction_db(afo_deesser *d, double v);
float afo_deesser_process_sample(afo_deesser *d, float input);
void afo_deesser_process_block(afo_deesser *d, float *buf, size_t n);
void afo_deesser_reset(afo_deesser *d);

/* ------------------------------------------------------------- prefilter */
/* audio/processor/routing.rs:9-12,826-843; processor.rs:74-76 */
typedef struct {
  float dc_x1, dc_y1;
  afo_biquad hp;
} afo_prefilter;
void afo_prefilter_init(afo_prefilter *p, double sample_rate);
void afo_prefilter_process_block(afo_prefilter *p, float *buf, size_t n, int apply_fixed_highpass);
/* routing.rs:802-823: returns the number of clipped samples */
uint64_t afo_sanitize_and_clamp(float *buf, size_t n);

/* ------------------------------------------------- offline block processor */
/* audio/processor/block_processor.rs:1-28 */
typedef struct {
  float input_sample_peak, output_sample_peak, true_peak_limiter_input_peak, output_true_peak;
  float limiter_peak_gain_reduction_db, true_peak_limiter_gain_reduction_db;
  uint64_t true_peak_limited_events;
  float compressor_gain_reduction_db, deesser_gain_reduction_db;
} afo_block_stats;

/* audio/processor/block_processor.rs:31-60 */
typedef struct {
  afo_deesser deesser;
  afo_eq eq;
  afo_compressor compressor;
  afo_limiter limiter;
  afo_tp_limiter tp_limiter;
  afo_tp_detector tp_detector;
  int deesser_enabled, eq_enabled, compressor_enabled, limiter_enabled, eq_before_deesser;
} afo_chain;

afo_chain *afo_chain_new(double sample_rate);
void afo_chain_free(afo_chain *c);
void afo_chain_set_deesser_enabled(afo_chain *c, int e);
void afo_chain_set_eq_enabled(afo_chain *c, int e);
void afo_chain_set_compressor_enabled(afo_chain *c, int e);
void afo_chain_set_limiter_enabled(afo_chain *c, int e);
void afo_chain_set_eq_before_deesser(afo_chain *c, int e);
afo_deesser *afo_chain_deesser(afo_chain *c);
afo_eq *afo_chain_eq(afo_chain *c);
afo_compressor *afo_chain_compressor(afo_chain *c);
afo_limiter *afo_chain_limiter(afo_chain *c);
afo_tp_limiter *afo_chain_tp_limiter(afo_chain *c);
/* processes `n` samples in place as ONE reference block and returns its stats */
afo_block_stats afo_chain_process_block(afo_chain *c, float *block, size_t n);

/* ------------------------------------------------ simulate_auto_eq_chain */
/* audio/processor/python_api.rs:415-487 defaults */
typedef struct {
  int32_t has_eq_bands_v2;
  afo_eq_band_config eq_bands_v2[AFO_NUM_BANDS];
  int32_t deesser_enabled, deesser_auto_enabled;
  double deesser_auto_amount, deesser_low_cut_hz, deesser_high_cut_hz, deesser_threshold_db;
  double deesser_ratio, deesser_attack_ms, deesser_release_ms, deesser_max_reduction_db;
  int32_t eq_before_deesser;
  int32_t compressor_enabled;
  double compressor_threshold_db, compressor_ratio, compressor_attack_ms, compressor_release_ms;
  double compressor_makeup_gain_db;
  int32_t compressor_adaptive_release;
  double compressor_base_release_ms;
  int32_t compressor_auto_makeup_enabled;
  double compressor_target_lufs;
  int32_t compressor_sidechain_highpass_enabled;
  int32_t limiter_enabled;
  double limiter_ceiling_db;
  int32_t limiter_careful_output_enabled;
  double limiter_lookahead_ms, limiter_release_ms;
} afo_sim_settings;

void afo_sim_settings_default(afo_sim_settings *s);

/* the numeric keys of the returned dict, python_api.rs:649-712 */
typedef struct {
  float input_sample_peak_db, input_rms_db, output_sample_peak_db, pre_limiter_true_peak_db;
  float output_true_peak_db, output_rms_db, limiter_effective_ceiling_db, sample_headroom_db;
  float pre_limiter_true_peak_headroom_db, true_peak_headroom_db, limiter_gain_reduction_db;
  float true_peak_limiter_gain_reduction_db;
  uint64_t true_peak_limited_events;
  float compressor_gain_reduction_db, deesser_gain_reduction_db;
  float compressor_gain_reduction_median_db, compressor_gain_reduction_p95_db;
  float compressor_gain_reduction_active_ratio, active_output_gain_db, silence_output_gain_db;
  float silence_level_delta_db, compressor_pumping_score_db;
  int32_t non_finite_output;
  float deesser_gain_reduction_median_db, deesser_gain_reduction_p95_db, analysis_block_ms;
  float active_analysis_threshold_db;
  uint64_t active_analysis_block_count, processed_samples;
} afo_sim_result;

/* returns 0 ok, -1 bad sample rate, -2 bad band.  out_audio may be NULL. */
int afo_simulate_auto_eq_chain(const float *audio, size_t n, double sample_rate,
                               const double bands[AFO_NUM_BANDS][3],
                               const afo_sim_settings *settings, afo_sim_result *result,
                               float *out_audio);

/* per-block rows -> dict statistics; python_api.rs:578-648 (shared with nothing in the product) */
float afo_percentile_f32(float *values, size_t n, float percentile);
float afo_pumping_score(const float *gr_trace_db, size_t n, float cadence_hz);
float afo_linear_to_db_f32(float v);

/* ---------------------------------------------------------- simulate_eq_v2 */
typedef struct {
  float input_sample_peak, output_sample_peak, input_true_peak, output_true_peak;
  double input_rms, output_rms, max_response_db;
  uint64_t sample_count;
  int32_t non_finite_output;
} afo_eq_v2_result;

int afo_simulate_eq_v2(const float *audio, size_t n, double sample_rate,
                       const afo_eq_band_config bands[AFO_NUM_BANDS], afo_eq_v2_result *result,
                       float *out_audio);
int afo_eq_magnitude_response(const double *freqs, size_t n, const double bands[AFO_NUM_BANDS][3],
                              double sample_rate, double *out);
int afo_eq_magnitude_response_v2(const double *freqs, size_t n,
                                 const afo_eq_band_config bands[AFO_NUM_BANDS], double sample_rate,
                                 double *out);

/* ------------------------------------------- simulate_auto_makeup_control */
/* python_api.rs:166-185 defaults */
typedef struct {
  double threshold_db, ratio, attack_ms, release_ms, makeup_gain_db, target_lufs, vad_reliability;
  int32_t adaptive_release, sidechain_highpass_enabled;
} afo_makeup_settings;
int afo_simulate_auto_makeup_control(const float *audio, size_t n, double sample_rate,
                                     const double *vad_probabilities, size_t n_vad,
                                     double noise_floor_db, double noise_reliability,
                                     const afo_makeup_settings *s, float *traces, float *out_audio);
int afo_measure_integrated_loudness(const float *audio, size_t n, uint32_t sample_rate, double *lufs);

/* ------------------------------------------------------------- utilities */
double afo_time_constant_to_coeff(double time_ms, double sample_rate);
double afo_db_to_linear(double db);
double afo_linear_to_db(double linear, double min_linear);
/* golden-KAT input generator, tests.rs:1824-1851 generalised per SURVEY 8(d) S3 */
void afo_kat_signal(float *out, size_t n_blocks, uint64_t noise_state0, double fundamental_hz,
                    double phrase_hz);


/* ---- noise gate (dsp/gate.rs): the downward-expander path that runs when no VadAutoGate is attached, which is
 * what simulate_gate_suppressor_order builds (python_api.rs:312-319: NoiseGate::new + set_gate_mode only).
 * `vad_mode` != 0 (VadAssisted / VadOnly) arms the chatter auto-relax (gate.rs:598-601). ---- */
typedef struct afo_gate {
  double threshold_db, attack_coeff, release_coeff, rms_coeff, sample_rate;
  double rms_envelope_sq, detector_level_db, current_gain;
  size_t hold_remaining_samples;
  int is_open, enabled, vad_mode;
  int effective_gate_open, has_effective_gate_state;
  size_t chatter_window_remaining_samples, chatter_cooldown_samples, auto_relax_remaining_samples;
  uint32_t chatter_transition_count;
  uint64_t chatter_event_count;
} afo_gate;
void afo_gate_init(afo_gate *g, double threshold_db, double attack_ms, double release_ms, double sample_rate);
void afo_gate_set_vad_mode(afo_gate *g, int vad_mode);
float afo_gate_process_sample(afo_gate *g, float input);
void afo_gate_process_block(afo_gate *g, float *buf, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* AF_ORACLE_H */
