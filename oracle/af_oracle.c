/*
 * af_oracle.c -- CPU restatement of AudioForge's rust-core voice chain.
 * TEST INFRASTRUCTURE ONLY (see af_oracle.h).  Build: -O2 -ffp-contract=off.
 *
 * Conventions used to mirror the Rust text:
 *   f64::max/min   -> fmax/fmin      (NaN-ignoring, like Rust)
 *   x.clamp(a,b)   -> clampd/clampf  (x<a?a : x>b?b : x ; NaN passes through)
 *   a.mul_add(b,c) -> fmaf(a,b,c)
 *   powf/exp/log10 -> libm
 *   `as f32`       -> (float) cast (round to nearest even)
 */
#include "af_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tp_fir_table.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static inline double clampd(double x, double lo, double hi) {
  if (x < lo) return lo;
  if (x > hi) return hi;
  return x;
}
static inline float clampf(float x, float lo, float hi) {
  if (x < lo) return lo;
  if (x > hi) return hi;
  return x;
}
static inline size_t clampz(size_t x, size_t lo, size_t hi) {
  return x < lo ? lo : (x > hi ? hi : x);
}

/* ------------------------------------------------------------------ util */
/* dsp/util.rs:5-8 */
double afo_time_constant_to_coeff(double time_ms, double sample_rate) {
  double tau = fmax(time_ms, 0.001) / 1000.0;
  return exp(-1.0 / (tau * sample_rate));
}
/* dsp/util.rs:12-14 */
double afo_db_to_linear(double db) { return pow(10.0, db / 20.0); }
/* dsp/util.rs:18-20 */
double afo_linear_to_db(double linear, double min_linear) {
  return 20.0 * log10(fmax(fabs(linear), min_linear));
}

/* ---------------------------------------------------------------- biquad */
#define MIN_BIQUAD_Q 1e-6
/* dsp/biquad.rs:12-19 */
static size_t coefficient_crossfade_samples(double sample_rate) {
  double samples = round(sample_rate * 1.5 / 1000.0);
  if (isfinite(samples)) {
    size_t s = samples <= 0.0 ? 0 : (size_t)samples;
    return clampz(s, 1, 4096);
  }
  return 1;
}

/* dsp/biquad.rs:110-182 */
void afo_biquad_coefficients(const afo_biquad *f, double out[5]) {
  double omega = 2.0 * M_PI * f->frequency / f->sample_rate;
  double sin_omega = sin(omega);
  double cos_omega = cos(omega);
  double q = fmax(f->q, MIN_BIQUAD_Q);
  double alpha = sin_omega / (2.0 * q);
  double a = pow(10.0, f->gain_db / 40.0);
  double b0, b1, b2, a0, a1, a2;
  switch (f->type) {
    case AFO_BQ_PEAKING:
      b0 = 1.0 + alpha * a;
      b1 = -2.0 * cos_omega;
      b2 = 1.0 - alpha * a;
      a0 = 1.0 + alpha / a;
      a1 = -2.0 * cos_omega;
      a2 = 1.0 - alpha / a;
      break;
    case AFO_BQ_LOWSHELF: {
      double t = 2.0 * sqrt(a) * alpha;
      b0 = a * ((a + 1.0) - (a - 1.0) * cos_omega + t);
      b1 = 2.0 * a * ((a - 1.0) - (a + 1.0) * cos_omega);
      b2 = a * ((a + 1.0) - (a - 1.0) * cos_omega - t);
      a0 = (a + 1.0) + (a - 1.0) * cos_omega + t;
      a1 = -2.0 * ((a - 1.0) + (a + 1.0) * cos_omega);
      a2 = (a + 1.0) + (a - 1.0) * cos_omega - t;
      break;
    }
    case AFO_BQ_HIGHSHELF: {
      double t = 2.0 * sqrt(a) * alpha;
      b0 = a * ((a + 1.0) + (a - 1.0) * cos_omega + t);
      b1 = -2.0 * a * ((a - 1.0) + (a + 1.0) * cos_omega);
      b2 = a * ((a + 1.0) + (a - 1.0) * cos_omega - t);
      a0 = (a + 1.0) - (a - 1.0) * cos_omega + t;
      a1 = 2.0 * ((a - 1.0) - (a + 1.0) * cos_omega);
      a2 = (a + 1.0) - (a - 1.0) * cos_omega - t;
      break;
    }
    case AFO_BQ_NOTCH:
      b0 = 1.0;
      b1 = -2.0 * cos_omega;
      b2 = 1.0;
      a0 = 1.0 + alpha;
      a1 = -2.0 * cos_omega;
      a2 = 1.0 - alpha;
      break;
    case AFO_BQ_HIGHPASS:
      b0 = (1.0 + cos_omega) / 2.0;
      b1 = -(1.0 + cos_omega);
      b2 = (1.0 + cos_omega) / 2.0;
      a0 = 1.0 + alpha;
      a1 = -2.0 * cos_omega;
      a2 = 1.0 - alpha;
      break;
    case AFO_BQ_LOWPASS:
      b0 = (1.0 - cos_omega) / 2.0;
      b1 = 1.0 - cos_omega;
      b2 = (1.0 - cos_omega) / 2.0;
      a0 = 1.0 + alpha;
      a1 = -2.0 * cos_omega;
      a2 = 1.0 - alpha;
      break;
    default: /* Bypass */
      b0 = 1.0; b1 = 0.0; b2 = 0.0; a0 = 1.0; a1 = 0.0; a2 = 0.0;
      break;
  }
  out[0] = b0 / a0;
  out[1] = b1 / a0;
  out[2] = b2 / a0;
  out[3] = a1 / a0;
  out[4] = a2 / a0;
}

/* dsp/biquad.rs:184-205 */
static double coefficient_magnitude_response_db(const double c[5], double frequency_hz,
                                                double sample_rate) {
  double omega = 2.0 * M_PI * frequency_hz / sample_rate;
  double cos_o = cos(omega), sin_o = sin(omega);
  double cos_2o = cos(2.0 * omega), sin_2o = sin(2.0 * omega);
  double nr = c[0] + c[1] * cos_o + c[2] * cos_2o;
  double ni = -c[1] * sin_o - c[2] * sin_2o;
  double dr = 1.0 + c[3] * cos_o + c[4] * cos_2o;
  double di = -c[3] * sin_o - c[4] * sin_2o;
  double np = nr * nr + ni * ni;
  double dp = dr * dr + di * di;
  double magnitude = sqrt(np / fmax(dp, 1.0e-30));
  return 20.0 * log10(fmax(magnitude, 1.0e-10));
}

/* dsp/biquad.rs:220-230 */
double afo_biquad_target_magnitude_db(const afo_biquad *f, double frequency_hz) {
  if (!f->enabled) return 0.0;
  double c[5];
  afo_biquad_coefficients(f, c);
  return coefficient_magnitude_response_db(c, frequency_hz, f->sample_rate);
}

/* dsp/biquad.rs:232-247 -- note: active z1/z2 are NOT touched */
static void set_coefficients_immediate(afo_biquad *f, const double c[5]) {
  f->b0 = c[0]; f->b1 = c[1]; f->b2 = c[2]; f->a1 = c[3]; f->a2 = c[4];
  f->pb0 = c[0]; f->pb1 = c[1]; f->pb2 = c[2]; f->pa1 = c[3]; f->pa2 = c[4];
  f->pz1 = 0.0;
  f->pz2 = 0.0;
  f->xf_total = 0;
  f->xf_remaining = 0;
}

/* dsp/biquad.rs:249-260 */
static void schedule_coefficients_crossfade(afo_biquad *f, const double c[5]) {
  f->pb0 = c[0]; f->pb1 = c[1]; f->pb2 = c[2]; f->pa1 = c[3]; f->pa2 = c[4];
  f->pz1 = f->z1;
  f->pz2 = f->z2;
  f->xf_total = coefficient_crossfade_samples(f->sample_rate);
  f->xf_remaining = f->xf_total;
}

/* dsp/biquad.rs:70-107 */
void afo_biquad_init(afo_biquad *f, afo_biquad_type type, double frequency, double gain_db,
                     double q, double sample_rate) {
  memset(f, 0, sizeof(*f));
  f->b0 = 1.0;
  f->pb0 = 1.0;
  f->type = type;
  f->frequency = frequency;
  f->gain_db = gain_db;
  f->q = q;
  f->sample_rate = sample_rate;
  f->enabled = 1;
  double c[5];
  afo_biquad_coefficients(f, c);
  set_coefficients_immediate(f, c);
}

/* dsp/biquad.rs:263-274 */
static inline double process_direct(double input, double b0, double b1, double b2, double a1,
                                    double a2, double *z1, double *z2) {
  double output = b0 * input + *z1;
  *z1 = b1 * input - a1 * output + *z2;
  *z2 = b2 * input - a2 * output;
  return output;
}

/* dsp/biquad.rs:276-286 */
static void promote_pending(afo_biquad *f) {
  f->b0 = f->pb0; f->b1 = f->pb1; f->b2 = f->pb2; f->a1 = f->pa1; f->a2 = f->pa2;
  f->z1 = f->pz1;
  f->z2 = f->pz2;
  f->xf_total = 0;
  f->xf_remaining = 0;
}

/* dsp/biquad.rs:290-327 */
float afo_biquad_process_sample(afo_biquad *f, float input) {
  if (!f->enabled) return input;
  double x = (double)input;
  double active = process_direct(x, f->b0, f->b1, f->b2, f->a1, f->a2, &f->z1, &f->z2);
  if (f->xf_remaining == 0) return (float)active;
  double pending = process_direct(x, f->pb0, f->pb1, f->pb2, f->pa1, f->pa2, &f->pz1, &f->pz2);
  size_t fade_pos = f->xf_total - f->xf_remaining + 1;
  double fade = (double)fade_pos / (double)f->xf_total;
  double output = active * (1.0 - fade) + pending * fade;
  f->xf_remaining -= 1;
  if (f->xf_remaining == 0) promote_pending(f);
  return (float)output;
}

/* dsp/biquad.rs:330-338 */
void afo_biquad_process_block(afo_biquad *f, float *buf, size_t n) {
  if (!f->enabled) return;
  for (size_t i = 0; i < n; ++i) buf[i] = afo_biquad_process_sample(f, buf[i]);
}

/* dsp/biquad.rs:341-347 */
void afo_biquad_reset(afo_biquad *f) {
  double c[5];
  afo_biquad_coefficients(f, c);
  set_coefficients_immediate(f, c);
}
/* dsp/biquad.rs:350-353 */
void afo_biquad_set_frequency(afo_biquad *f, double frequency) {
  f->frequency = frequency;
  double c[5];
  afo_biquad_coefficients(f, c);
  schedule_coefficients_crossfade(f, c);
}
/* dsp/biquad.rs:356-359 */
void afo_biquad_set_gain_db(afo_biquad *f, double gain_db) {
  f->gain_db = gain_db;
  double c[5];
  afo_biquad_coefficients(f, c);
  schedule_coefficients_crossfade(f, c);
}
/* dsp/biquad.rs:365-368 */
void afo_biquad_set_gain_db_immediate(afo_biquad *f, double gain_db) {
  f->gain_db = gain_db;
  double c[5];
  afo_biquad_coefficients(f, c);
  set_coefficients_immediate(f, c);
}
/* dsp/biquad.rs:371-374 */
void afo_biquad_set_q(afo_biquad *f, double q) {
  f->q = fmax(q, MIN_BIQUAD_Q);
  double c[5];
  afo_biquad_coefficients(f, c);
  schedule_coefficients_crossfade(f, c);
}
/* dsp/biquad.rs:377-389 */
void afo_biquad_set_parameters(afo_biquad *f, afo_biquad_type type, double frequency,
                               double gain_db, double q) {
  f->type = type;
  f->frequency = frequency;
  f->gain_db = gain_db;
  f->q = fmax(q, MIN_BIQUAD_Q);
  double c[5];
  afo_biquad_coefficients(f, c);
  schedule_coefficients_crossfade(f, c);
}
/* dsp/biquad.rs:395-407 */
void afo_biquad_set_parameters_immediate(afo_biquad *f, afo_biquad_type type, double frequency,
                                         double gain_db, double q) {
  f->type = type;
  f->frequency = frequency;
  f->gain_db = gain_db;
  f->q = fmax(q, MIN_BIQUAD_Q);
  double c[5];
  afo_biquad_coefficients(f, c);
  set_coefficients_immediate(f, c);
}

/* -------------------------------------------------------------------- EQ */
static const double DEFAULT_FREQUENCIES[AFO_NUM_BANDS] = {80.0,   160.0,  320.0,  640.0,   1280.0,
                                                          2500.0, 5000.0, 8000.0, 12000.0, 16000.0};
#define DEFAULT_Q 1.41

/* dsp/eq.rs:122-138 */
static afo_eq_band_config eq_default_config(size_t index) {
  afo_eq_band_config c;
  c.filter_type = index == 0 ? AFO_EQ_LOW_SHELF : (index == 9 ? AFO_EQ_HIGH_SHELF : AFO_EQ_BELL);
  c.frequency_hz = DEFAULT_FREQUENCIES[index];
  c.gain_db = 0.0;
  c.q = DEFAULT_Q;
  c.slope_db_per_octave = 12;
  c.enabled = 1;
  return c;
}

static int eq_is_pass(int32_t t) { return t == AFO_EQ_HIGH_PASS || t == AFO_EQ_LOW_PASS; }

/* dsp/eq.rs:97-107 */
static afo_biquad_type eq_biquad_type(int32_t t) {
  switch (t) {
    case AFO_EQ_LOW_SHELF: return AFO_BQ_LOWSHELF;
    case AFO_EQ_BELL: return AFO_BQ_PEAKING;
    case AFO_EQ_HIGH_SHELF: return AFO_BQ_HIGHSHELF;
    case AFO_EQ_NOTCH: return AFO_BQ_NOTCH;
    case AFO_EQ_HIGH_PASS: return AFO_BQ_HIGHPASS;
    default: return AFO_BQ_LOWPASS;
  }
}

/* dsp/eq.rs:203-207 */
static double butterworth_section_q(size_t section_index, size_t section_count) {
  size_t order = 2 * section_count;
  double angle = (double)(2 * section_index + 1) * M_PI / (double)(2 * order);
  return 1.0 / (2.0 * cos(angle));
}

static int slope_supported(int32_t s) { return s == 12 || s == 24 || s == 36 || s == 48; }

/* dsp/eq.rs:248-256 */
static size_t eq_required_sections(const afo_eq_band_config *c) {
  if (!c->enabled) return 0;
  if (eq_is_pass(c->filter_type))
    return slope_supported(c->slope_db_per_octave) ? (size_t)c->slope_db_per_octave / 12 : 1;
  return 1;
}

/* dsp/eq.rs:258-277 */
static void eq_section_parameters(const afo_eq_band_config *c, size_t index, size_t count,
                                  afo_biquad_type *type, double *gain_db, double *q) {
  if (eq_is_pass(c->filter_type)) {
    *type = eq_biquad_type(c->filter_type);
    *gain_db = 0.0;
    *q = butterworth_section_q(index, count);
  } else {
    *type = eq_biquad_type(c->filter_type);
    *gain_db = c->filter_type == AFO_EQ_NOTCH ? 0.0 : c->gain_db;
    *q = c->q;
  }
}

/* dsp/eq.rs:223-246 */
static void eq_band_init(afo_eq_band *b, afo_eq_band_config config, double sample_rate) {
  size_t target = eq_required_sections(&config);
  for (size_t s = 0; s < AFO_MAX_PASS_SECTIONS; ++s) {
    if (s < target) {
      afo_biquad_type t;
      double g, q;
      eq_section_parameters(&config, s, target, &t, &g, &q);
      afo_biquad_init(&b->sections[s], t, config.frequency_hz, g, q, sample_rate);
    } else {
      afo_biquad_init(&b->sections[s], AFO_BQ_BYPASS, config.frequency_hz, 0.0, DEFAULT_Q,
                      sample_rate);
    }
  }
  b->config = config;
  b->processing_sections = target;
  b->target_sections = target;
}

/* dsp/eq.rs:279-298 */
static void eq_band_set_config(afo_eq_band *b, afo_eq_band_config config) {
  b->config = config;
  size_t target = eq_required_sections(&config);
  size_t processing = b->processing_sections > target ? b->processing_sections : target;
  for (size_t s = 0; s < processing; ++s) {
    afo_biquad_type t = AFO_BQ_BYPASS;
    double g = 0.0, q = DEFAULT_Q;
    if (s < target) eq_section_parameters(&config, s, target, &t, &g, &q);
    afo_biquad_set_parameters(&b->sections[s], t, config.frequency_hz, g, q);
  }
  b->processing_sections = processing;
  b->target_sections = target;
}

/* dsp/eq.rs:300-306 */
static void eq_band_finish_retired(afo_eq_band *b) {
  while (b->processing_sections > b->target_sections &&
         b->sections[b->processing_sections - 1].xf_remaining == 0) {
    b->processing_sections -= 1;
  }
}

/* dsp/eq.rs:317-322 */
static void eq_band_process_block(afo_eq_band *b, float *buf, size_t n) {
  for (size_t s = 0; s < b->processing_sections; ++s) afo_biquad_process_block(&b->sections[s], buf, n);
  eq_band_finish_retired(b);
}

/* dsp/eq.rs:324-336 */
static void eq_band_reset(afo_eq_band *b) {
  size_t target = eq_required_sections(&b->config);
  for (size_t s = 0; s < AFO_MAX_PASS_SECTIONS; ++s) {
    afo_biquad_type t = AFO_BQ_BYPASS;
    double g = 0.0, q = DEFAULT_Q;
    if (s < target) eq_section_parameters(&b->config, s, target, &t, &g, &q);
    afo_biquad_set_parameters_immediate(&b->sections[s], t, b->config.frequency_hz, g, q);
  }
  b->processing_sections = target;
  b->target_sections = target;
}

/* dsp/eq.rs:357-368 */
void afo_eq_init(afo_eq *eq, double sample_rate) {
  for (size_t i = 0; i < AFO_NUM_BANDS; ++i) eq_band_init(&eq->bands[i], eq_default_config(i), sample_rate);
  eq->enabled = 1;
  eq->sample_rate = sample_rate;
}
/* dsp/eq.rs:371-379 -- band-major over the block */
void afo_eq_process_block(afo_eq *eq, float *buf, size_t n) {
  if (!eq->enabled) return;
  for (size_t i = 0; i < AFO_NUM_BANDS; ++i) eq_band_process_block(&eq->bands[i], buf, n);
}
/* dsp/eq.rs:395-399 */
void afo_eq_reset(afo_eq *eq) {
  for (size_t i = 0; i < AFO_NUM_BANDS; ++i) eq_band_reset(&eq->bands[i]);
}
/* dsp/eq.rs:406-412 */
void afo_eq_set_band_gain(afo_eq *eq, size_t band, double gain_db) {
  if (band >= AFO_NUM_BANDS) return;
  afo_eq_band_config c = eq->bands[band].config;
  c.gain_db = gain_db;
  eq_band_set_config(&eq->bands[band], c);
}
/* dsp/eq.rs:419-425 */
void afo_eq_set_band_frequency(afo_eq *eq, size_t band, double frequency) {
  if (band >= AFO_NUM_BANDS) return;
  afo_eq_band_config c = eq->bands[band].config;
  c.frequency_hz = frequency;
  eq_band_set_config(&eq->bands[band], c);
}
/* dsp/eq.rs:432-438 */
void afo_eq_set_band_q(afo_eq *eq, size_t band, double q) {
  if (band >= AFO_NUM_BANDS) return;
  afo_eq_band_config c = eq->bands[band].config;
  c.q = q;
  eq_band_set_config(&eq->bands[band], c);
}
/* dsp/eq.rs:468-472 */
void afo_eq_set_band_config(afo_eq *eq, size_t band, const afo_eq_band_config *config) {
  if (band >= AFO_NUM_BANDS) return;
  eq_band_set_config(&eq->bands[band], *config);
}
/* dsp/eq.rs:338-343, 511-527 */
void afo_eq_magnitude_response_db(const afo_eq *eq, const double *freqs, size_t n, double *out) {
  for (size_t i = 0; i < n; ++i) {
    if (!eq->enabled) { out[i] = 0.0; continue; }
    double total = 0.0;
    for (size_t b = 0; b < AFO_NUM_BANDS; ++b) {
      double band_sum = 0.0;
      for (size_t s = 0; s < eq->bands[b].target_sections; ++s)
        band_sum += afo_biquad_target_magnitude_db(&eq->bands[b].sections[s], freqs[i]);
      total += band_sum;
    }
    out[i] = total;
  }
}

/* dsp/eq.rs:140-201 */
int afo_eq_band_config_validate(const afo_eq_band_config *c, size_t index, double sample_rate,
                                char *msg, size_t msg_len) {
  const double fmin = 20.0;
  if (!isfinite(c->frequency_hz)) {
    snprintf(msg, msg_len, "Band %zu: frequency must be finite", index);
    return -1;
  }
  if (!isfinite(sample_rate) || sample_rate <= 2.0 * fmin) {
    snprintf(msg, msg_len, "Band %zu: sample rate must be finite and support the EQ frequency range",
             index);
    return -1;
  }
  double max_frequency = fmax(sample_rate / 2.0 - 1.0, fmin);
  if (!(c->frequency_hz >= fmin && c->frequency_hz <= max_frequency)) {
    snprintf(msg, msg_len, "Band %zu: frequency %g Hz out of range [%g, %g]", index,
             c->frequency_hz, fmin, max_frequency);
    return -1;
  }
  if (!isfinite(c->gain_db)) {
    snprintf(msg, msg_len, "Band %zu: gain must be finite", index);
    return -1;
  }
  if (!(c->gain_db >= -12.0 && c->gain_db <= 12.0)) {
    snprintf(msg, msg_len, "Band %zu: gain %g dB out of range [-12, 12]", index, c->gain_db);
    return -1;
  }
  if (!isfinite(c->q)) {
    snprintf(msg, msg_len, "Band %zu: Q must be finite", index);
    return -1;
  }
  if (!(c->q >= 0.1 && c->q <= 10.0)) {
    snprintf(msg, msg_len, "Band %zu: Q %g out of range [0.1, 10]", index, c->q);
    return -1;
  }
  if (!slope_supported(c->slope_db_per_octave)) {
    snprintf(msg, msg_len,
             "Band %zu: slope %d dB/octave is unsupported; expected one of [12, 24, 36, 48]", index,
             (int)c->slope_db_per_octave);
    return -1;
  }
  return 0;
}

/* -------------------------------------------------------------- loudness */
/*
 * ebur128 0.1.10 is not vendored in the reference; this follows the published
 * libebur128 design the crate ports (ITU-R BS.1770-4): K-weighting as one
 * 4th-order direct-form filter built from the shelving + RLB prototypes, and
 * momentary loudness = -0.691 + 10 log10(mean square over the last 400 ms).
 * Parity for this block is UNPINNED.
 */
static void kweight_design(double fs, double b[5], double a[5]) {
  double f0 = 1681.974450955533;
  double G = 3.999843853973347;
  double Q = 0.7071752369554196;
  double K = tan(M_PI * f0 / fs);
  double Vh = pow(10.0, G / 20.0);
  double Vb = pow(Vh, 0.4996667741545416);
  double pb[3], pa[3] = {1.0, 0.0, 0.0};
  double rb[3] = {1.0, -2.0, 1.0}, ra[3] = {1.0, 0.0, 0.0};
  double a0 = 1.0 + K / Q + K * K;
  pb[0] = (Vh + Vb * K / Q + K * K) / a0;
  pb[1] = 2.0 * (K * K - Vh) / a0;
  pb[2] = (Vh - Vb * K / Q + K * K) / a0;
  pa[1] = 2.0 * (K * K - 1.0) / a0;
  pa[2] = (1.0 - K / Q + K * K) / a0;
  f0 = 38.13547087602444;
  Q = 0.5003270373238773;
  K = tan(M_PI * f0 / fs);
  ra[1] = 2.0 * (K * K - 1.0) / (1.0 + K / Q + K * K);
  ra[2] = (1.0 - K / Q + K * K) / (1.0 + K / Q + K * K);
  b[0] = pb[0] * rb[0];
  b[1] = pb[0] * rb[1] + pb[1] * rb[0];
  b[2] = pb[0] * rb[2] + pb[1] * rb[1] + pb[2] * rb[0];
  b[3] = pb[1] * rb[2] + pb[2] * rb[1];
  b[4] = pb[2] * rb[2];
  a[0] = pa[0] * ra[0];
  a[1] = pa[0] * ra[1] + pa[1] * ra[0];
  a[2] = pa[0] * ra[2] + pa[1] * ra[1] + pa[2] * ra[0];
  a[3] = pa[1] * ra[2] + pa[2] * ra[1];
  a[4] = pa[2] * ra[2];
}

/* dsp/loudness.rs:36-41,99-113 */
int afo_loudness_init(afo_loudness *m, uint32_t sample_rate) {
  static const uint32_t ok[] = {8000, 16000, 32000, 44100, 48000, 88200, 96000};
  memset(m, 0, sizeof(*m));
  int found = 0;
  for (size_t i = 0; i < sizeof(ok) / sizeof(ok[0]); ++i) found |= (ok[i] == sample_rate);
  if (!found) return -1;
  kweight_design((double)sample_rate, m->b, m->a);
  size_t s100 = (sample_rate + 5) / 10;
  m->ring_frames = s100 * 4;
  m->ring = (double *)calloc(m->ring_frames, sizeof(double));
  if (!m->ring) return -1;
  m->ring_index = 0;
  m->current_lufs = -100.0f;
  m->sample_rate = sample_rate;
  m->valid = 1;
  return 0;
}
void afo_loudness_free(afo_loudness *m) {
  free(m->ring);
  m->ring = NULL;
  m->valid = 0;
}
/* dsp/loudness.rs:119-135 */
void afo_loudness_process(afo_loudness *m, const float *samples, size_t n) {
  if (!m->valid) return;
  for (size_t i = 0; i < n; ++i) {
    double *v = m->v;
    v[0] = (double)samples[i] - m->a[1] * v[1] - m->a[2] * v[2] - m->a[3] * v[3] - m->a[4] * v[4];
    double y = m->b[0] * v[0] + m->b[1] * v[1] + m->b[2] * v[2] + m->b[3] * v[3] + m->b[4] * v[4];
    v[4] = v[3];
    v[3] = v[2];
    v[2] = v[1];
    v[1] = v[0];
    m->ring[m->ring_index] = y;
    m->ring_index = (m->ring_index + 1) % m->ring_frames;
  }
  for (int k = 1; k <= 4; ++k)
    if (fabs(m->v[k]) < 2.2250738585072014e-308) m->v[k] = 0.0;
  double sum = 0.0;
  for (size_t i = 0; i < m->ring_frames; ++i) sum += m->ring[i] * m->ring[i];
  double energy = sum / (double)m->ring_frames;
  double lufs = energy <= 0.0 ? -HUGE_VAL : 10.0 * (log(energy) / log(10.0)) - 0.691;
  m->current_lufs = (float)lufs;
}
/* dsp/loudness.rs:149-157 */
void afo_loudness_reset(afo_loudness *m) {
  if (!m->valid) return;
  memset(m->v, 0, sizeof(m->v));
  memset(m->ring, 0, m->ring_frames * sizeof(double));
  m->ring_index = 0;
  m->current_lufs = -100.0f;
}

/* ------------------------------------------------------------ compressor */
#define DETECTOR_PEAK_WEIGHT 0.6
#define DETECTOR_RMS_WEIGHT 0.4
#define ADAPTIVE_FAST_RELEASE_MS 50.0
#define ADAPTIVE_SLOW_CHARGE_MS 250.0
#define ADAPTIVE_SLOW_RELEASE_MS 400.0
#define SLOW_RELEASE_TRIGGER_DB 3.0
#define SPEECH_ACTIVE_RMS_MIN_DB (-55.0)
#define SPEECH_ACTIVE_RMS_MAX_DB (-6.0)
#define AUTO_MAKEUP_ACTIVE_MIN 0.20
#define AUTO_MAKEUP_RELIABILITY_MIN 0.35
#define AUTO_MAKEUP_ACTIVITY_SMOOTH_MS 200.0
#define NOISE_RELATIVE_ACTIVITY_START_DB 3.0
#define NOISE_RELATIVE_ACTIVITY_FULL_DB 15.0
#define MAKEUP_SILENCE_RELAX_MS 1500.0
#define SIDECHAIN_HIGHPASS_DEFAULT_HZ 120.0
#define SIDECHAIN_BAND_ENV_MS 18.0
#define PLOSIVE_RATIO_START 1.25
#define PLOSIVE_RATIO_FULL 5.0
#define PLOSIVE_MIN_DETECTOR_GAIN 0.35

/* dsp/compressor.rs:390-394 */
static double sidechain_highpass_coeff(double cutoff_hz, double sample_rate) {
  cutoff_hz = clampd(cutoff_hz, 20.0, sample_rate * 0.45);
  double omega = 2.0 * M_PI * cutoff_hz / fmax(sample_rate, 1.0);
  return 1.0 / (1.0 + omega);
}

/* dsp/compressor.rs:133-202 */
void afo_compressor_init(afo_compressor *c, double threshold_db, double ratio, double attack_ms,
                         double release_ms, double makeup_gain_db, double knee_db,
                         double sample_rate) {
  memset(c, 0, sizeof(*c));
  double release_coeff = afo_time_constant_to_coeff(release_ms, sample_rate);
  c->threshold_db = threshold_db;
  c->ratio = fmax(ratio, 1.0);
  c->attack_coeff = afo_time_constant_to_coeff(attack_ms, sample_rate);
  c->release_coeff = release_coeff;
  c->detector_release_coeff = release_coeff;
  c->makeup_gain_db = makeup_gain_db;
  c->makeup_gain_linear = afo_db_to_linear(makeup_gain_db);
  c->knee_db = fmax(knee_db, 0.0);
  c->peak_envelope_db = -120.0;
  c->rms_envelope_sq = 0.0;
  c->rms_coeff = afo_time_constant_to_coeff(20.0, sample_rate);
  c->current_gain_reduction_db = 0.0;
  c->sample_rate = sample_rate;
  c->enabled = 1;
  c->adaptive_release = 0;
  c->base_release_ms = release_ms;
  c->current_release_ms = release_ms;
  c->target_release_ms = release_ms;
  c->release_smoothing_coeff = afo_time_constant_to_coeff(100.0, sample_rate);
  c->has_meter = afo_loudness_init(&c->meter, (uint32_t)sample_rate) == 0;
  c->auto_makeup_enabled = 0;
  c->target_lufs = -18.0;
  c->smoothed_makeup_gain = makeup_gain_db;
  c->makeup_smoothing_coeff = afo_time_constant_to_coeff(200.0, sample_rate);
  c->current_lufs = -100.0;
  c->speech_activity_smoothing_coeff =
      afo_time_constant_to_coeff(AUTO_MAKEUP_ACTIVITY_SMOOTH_MS, sample_rate);
  c->makeup_silence_relax_coeff = afo_time_constant_to_coeff(MAKEUP_SILENCE_RELAX_MS, sample_rate);
  c->sidechain_highpass_enabled = 0;
  c->sidechain_highpass_coeff = sidechain_highpass_coeff(SIDECHAIN_HIGHPASS_DEFAULT_HZ, sample_rate);
}
void afo_compressor_free(afo_compressor *c) {
  if (c->has_meter) afo_loudness_free(&c->meter);
  c->has_meter = 0;
}

/* dsp/compressor.rs:288-291 */
static void reset_adaptive_release_state(afo_compressor *c) {
  c->fast_release_env_db = c->current_gain_reduction_db;
  c->slow_release_env_db = 0.0;
}
/* dsp/compressor.rs:210-213 */
void afo_compressor_set_threshold(afo_compressor *c, double v) {
  c->threshold_db = v;
  reset_adaptive_release_state(c);
}
/* dsp/compressor.rs:221-223 */
void afo_compressor_set_ratio(afo_compressor *c, double v) { c->ratio = fmax(v, 1.0); }
/* dsp/compressor.rs:231-233 */
void afo_compressor_set_attack_time(afo_compressor *c, double ms) {
  c->attack_coeff = afo_time_constant_to_coeff(ms, c->sample_rate);
}
/* dsp/compressor.rs:236-244 */
void afo_compressor_set_release_time(afo_compressor *c, double ms) {
  c->base_release_ms = ms;
  if (!c->adaptive_release) {
    c->current_release_ms = ms;
    c->target_release_ms = ms;
    c->release_coeff = afo_time_constant_to_coeff(ms, c->sample_rate);
  }
  c->detector_release_coeff = afo_time_constant_to_coeff(ms, c->sample_rate);
}
/* dsp/compressor.rs:247-260 */
void afo_compressor_set_adaptive_release(afo_compressor *c, int enabled) {
  c->adaptive_release = enabled;
  if (!enabled) {
    c->current_release_ms = c->base_release_ms;
    c->target_release_ms = c->base_release_ms;
    c->fast_release_env_db = c->current_gain_reduction_db;
    c->slow_release_env_db = 0.0;
    c->release_coeff = afo_time_constant_to_coeff(c->current_release_ms, c->sample_rate);
  } else {
    c->fast_release_env_db = c->current_gain_reduction_db;
    c->slow_release_env_db = 0.0;
  }
}
/* dsp/compressor.rs:268-275 */
void afo_compressor_set_base_release_time(afo_compressor *c, double ms) {
  c->base_release_ms = ms;
  if (!c->adaptive_release) {
    c->current_release_ms = ms;
    c->target_release_ms = ms;
    c->release_coeff = afo_time_constant_to_coeff(ms, c->sample_rate);
  }
}
/* dsp/compressor.rs:294-300 */
void afo_compressor_set_makeup_gain(afo_compressor *c, double db) {
  c->makeup_gain_db = db;
  c->makeup_gain_linear = afo_db_to_linear(db);
  if (!c->auto_makeup_enabled) c->smoothed_makeup_gain = db;
}
/* dsp/compressor.rs:303-305 */
void afo_compressor_set_enabled(afo_compressor *c, int enabled) { c->enabled = enabled; }
/* dsp/compressor.rs:318-323 */
void afo_compressor_set_auto_makeup_enabled(afo_compressor *c, int enabled) {
  c->auto_makeup_enabled = enabled && c->has_meter;
  if (!enabled) c->smoothed_makeup_gain = c->makeup_gain_db;
}
/* dsp/compressor.rs:331-333 */
void afo_compressor_set_target_lufs(afo_compressor *c, double v) {
  c->target_lufs = clampd(v, -24.0, -12.0);
}
/* dsp/compressor.rs:516-518 */
static int finite_unit(double value, double *out) {
  if (!isfinite(value)) return 0;
  *out = clampd(value, 0.0, 1.0);
  return 1;
}
/* dsp/compressor.rs:351-353 */
void afo_compressor_set_noise_reference_reliability(afo_compressor *c, double v) {
  double u;
  c->noise_reference_reliability = finite_unit(v, &u) ? u : 0.0;
}
/* dsp/compressor.rs:397-404 */
static void reset_sidechain_highpass_state(afo_compressor *c) {
  c->sidechain_highpass_prev_input = 0.0;
  c->sidechain_highpass_prev_output = 0.0;
  c->low_band_env_sq = 0.0;
  c->voiced_band_env_sq = 0.0;
  c->presence_band_env_sq = 0.0;
  c->plosive_ratio = 0.0;
}
/* dsp/compressor.rs:366-371 */
void afo_compressor_set_sidechain_highpass_enabled(afo_compressor *c, int enabled) {
  if ((c->sidechain_highpass_enabled != 0) != (enabled != 0)) reset_sidechain_highpass_state(c);
  c->sidechain_highpass_enabled = enabled;
}
/* dsp/compressor.rs:385-387 */
void afo_compressor_set_limiter_feedback_gain_reduction_db(afo_compressor *c, double v) {
  c->limiter_feedback_gain_reduction_db = clampd(v, 0.0, 24.0);
}

/* dsp/compressor.rs:407-417 */
static inline double process_sidechain_sample(afo_compressor *c, double input) {
  if (!c->sidechain_highpass_enabled) return input;
  double output = c->sidechain_highpass_coeff *
                  (c->sidechain_highpass_prev_output + input - c->sidechain_highpass_prev_input);
  c->sidechain_highpass_prev_input = input;
  c->sidechain_highpass_prev_output = output;
  return output;
}

/* dsp/compressor.rs:420-450 */
static inline double update_sidechain_band_metrics(afo_compressor *c, double full_band_input,
                                                   double detector_input) {
  if (!c->sidechain_highpass_enabled) {
    c->plosive_ratio = 0.0;
    return 1.0;
  }
  double low_component = full_band_input - detector_input;
  double voiced_component = detector_input;
  double presence_component = 0.65 * detector_input + 0.35 * (detector_input - low_component);
  double coeff = afo_time_constant_to_coeff(SIDECHAIN_BAND_ENV_MS, c->sample_rate);

  c->low_band_env_sq = coeff * c->low_band_env_sq + (1.0 - coeff) * low_component * low_component;
  c->voiced_band_env_sq =
      coeff * c->voiced_band_env_sq + (1.0 - coeff) * voiced_component * voiced_component;
  c->presence_band_env_sq =
      coeff * c->presence_band_env_sq + (1.0 - coeff) * presence_component * presence_component;

  double low_rms = sqrt(c->low_band_env_sq);
  double voiced_rms = fmax(sqrt(c->voiced_band_env_sq), 1e-8);
  double presence_rms = sqrt(c->presence_band_env_sq);
  c->plosive_ratio = clampd(low_rms / voiced_rms, 0.0, 32.0);

  double plosive_amount =
      clampd((c->plosive_ratio - PLOSIVE_RATIO_START) / (PLOSIVE_RATIO_FULL - PLOSIVE_RATIO_START),
             0.0, 1.0);
  double plosive_penalty = 1.0 - plosive_amount * (1.0 - PLOSIVE_MIN_DETECTOR_GAIN);
  double presence_ratio = clampd(presence_rms / voiced_rms, 0.0, 4.0);
  double presence_weight = 1.0 + 0.18 * clampd(presence_ratio - 0.75, 0.0, 1.0);
  return clampd(plosive_penalty * presence_weight, PLOSIVE_MIN_DETECTOR_GAIN, 1.15);
}

/* dsp/compressor.rs:452-466 */
static inline void update_adaptive_release_time_meter(afo_compressor *c) {
  if (!c->adaptive_release) {
    c->target_release_ms = c->base_release_ms;
    return;
  }
  double sustained = clampd(c->slow_release_env_db / (SLOW_RELEASE_TRIGGER_DB + 3.0), 0.0, 1.0);
  double transient_bias = clampd(
      (c->fast_release_env_db - c->slow_release_env_db) / (SLOW_RELEASE_TRIGGER_DB + 4.0), 0.0, 1.0);
  double syllabic = clampd(sustained * sustained * (1.0 - 0.35 * transient_bias), 0.0, 1.0);
  c->target_release_ms =
      ADAPTIVE_FAST_RELEASE_MS + syllabic * (ADAPTIVE_SLOW_RELEASE_MS - ADAPTIVE_FAST_RELEASE_MS);
}

/* dsp/compressor.rs:468-505 */
static inline void smooth_gain_reduction(afo_compressor *c, double target) {
  if (!c->adaptive_release) {
    double k = target > c->current_gain_reduction_db ? c->attack_coeff : c->release_coeff;
    c->current_gain_reduction_db = k * c->current_gain_reduction_db + (1.0 - k) * target;
    c->fast_release_env_db = c->current_gain_reduction_db;
    c->slow_release_env_db = 0.0;
    return;
  }
  double fast_release_coeff = afo_time_constant_to_coeff(ADAPTIVE_FAST_RELEASE_MS, c->sample_rate);
  double slow_charge_coeff = afo_time_constant_to_coeff(ADAPTIVE_SLOW_CHARGE_MS, c->sample_rate);
  double slow_release_coeff = afo_time_constant_to_coeff(ADAPTIVE_SLOW_RELEASE_MS, c->sample_rate);

  if (target > c->current_gain_reduction_db) {
    c->fast_release_env_db =
        c->attack_coeff * c->current_gain_reduction_db + (1.0 - c->attack_coeff) * target;
  } else {
    c->fast_release_env_db =
        fast_release_coeff * c->fast_release_env_db + (1.0 - fast_release_coeff) * target;
  }
  if (target > SLOW_RELEASE_TRIGGER_DB) {
    c->slow_release_env_db =
        slow_charge_coeff * c->slow_release_env_db + (1.0 - slow_charge_coeff) * target;
  } else {
    c->slow_release_env_db *= slow_release_coeff;
  }
  c->current_gain_reduction_db = fmax(c->fast_release_env_db, c->slow_release_env_db);
}

/* dsp/compressor.rs:507-514 */
static double speech_activity_from_rms_db(double rms_db) {
  if (!(rms_db >= SPEECH_ACTIVE_RMS_MIN_DB && rms_db <= SPEECH_ACTIVE_RMS_MAX_DB)) return 0.0;
  double onset = clampd((rms_db - SPEECH_ACTIVE_RMS_MIN_DB) / 12.0, 0.0, 1.0);
  double overload = clampd((SPEECH_ACTIVE_RMS_MAX_DB - rms_db) / 6.0, 0.0, 1.0);
  return fmin(onset, overload);
}
/* dsp/compressor.rs:520-526 */
static double smoothstep(double edge0, double edge1, double value) {
  if (!isfinite(value) || !isfinite(edge0) || !isfinite(edge1) || edge1 <= edge0) return 0.0;
  double t = clampd((value - edge0) / (edge1 - edge0), 0.0, 1.0);
  return t * t * (3.0 - 2.0 * t);
}

/* dsp/compressor.rs:528-581 */
static void estimate_auto_makeup_activity(const afo_compressor *c, double rms_db,
                                          const afo_auto_makeup_input *ev, double *activity,
                                          double *reliability) {
  double absolute_activity = speech_activity_from_rms_db(rms_db);
  if (!ev) {
    *activity = absolute_activity;
    *reliability = 1.0;
    return;
  }
  double vad_reliability = 0.0, vad_probability = 0.0, u;
  if (finite_unit(ev->vad_reliability, &u)) vad_reliability = u;
  if (finite_unit(ev->vad_probability, &u)) {
    vad_probability = u;
  } else {
    vad_reliability = 0.0;
    vad_probability = 0.0;
  }
  double configured = finite_unit(c->noise_reference_reliability, &u) ? u : 0.0;
  double live = finite_unit(ev->live_noise_reliability, &u) ? u : 0.0;
  double noise_reliability = configured > 0.0 ? fmin(live, configured) : live;
  double relative_activity;
  if (isfinite(ev->noise_floor_db) && ev->noise_floor_db >= -120.0 && ev->noise_floor_db <= 0.0) {
    relative_activity = smoothstep(ev->noise_floor_db + NOISE_RELATIVE_ACTIVITY_START_DB,
                                   ev->noise_floor_db + NOISE_RELATIVE_ACTIVITY_FULL_DB, rms_db);
  } else {
    noise_reliability = 0.0;
    relative_activity = 0.0;
  }
  double fallback = noise_reliability * relative_activity + (1.0 - noise_reliability) * absolute_activity;
  double act = vad_reliability * vad_probability + (1.0 - vad_reliability) * fallback;
  double rel = fmax(vad_reliability, 0.75 * noise_reliability);
  *activity = clampd(act, 0.0, 1.0);
  *reliability = clampd(rel, 0.0, 1.0);
}

/* dsp/compressor.rs:583-596 */
static double block_rms_db(const float *buf, size_t n) {
  if (n == 0) return -120.0;
  double sum = 0.0;
  for (size_t i = 0; i < n; ++i) {
    double s = (double)buf[i];
    sum += s * s;
  }
  double power = sum / (double)n;
  return afo_linear_to_db(sqrt(power), 1e-10);
}

/* dsp/compressor.rs:598-653 */
static void update_auto_makeup_gain(afo_compressor *c, double speech_activity, double reliability,
                                    size_t elapsed_samples) {
  double elapsed = (double)(elapsed_samples < 1 ? 1 : elapsed_samples);
  double makeup_coeff = pow(c->makeup_smoothing_coeff, elapsed);
  double silence_relax_coeff = pow(c->makeup_silence_relax_coeff, elapsed);
  if (!c->auto_makeup_enabled) {
    double target = c->makeup_gain_db;
    double diff = target - c->smoothed_makeup_gain;
    if (fabs(diff) > 0.1) {
      c->smoothed_makeup_gain = makeup_coeff * c->smoothed_makeup_gain + (1.0 - makeup_coeff) * target;
    } else {
      c->smoothed_makeup_gain = target;
    }
    return;
  }
  if (c->has_meter) {
    c->current_lufs = (double)c->meter.current_lufs;
    double activity_coeff = pow(c->speech_activity_smoothing_coeff, elapsed);
    c->speech_activity_score = activity_coeff * c->speech_activity_score +
                               (1.0 - activity_coeff) * clampd(speech_activity, 0.0, 1.0);
    c->auto_makeup_activity_reliability = clampd(reliability, 0.0, 1.0);
    if (c->speech_activity_score < AUTO_MAKEUP_ACTIVE_MIN) {
      c->smoothed_makeup_gain = silence_relax_coeff * c->smoothed_makeup_gain +
                                (1.0 - silence_relax_coeff) * c->makeup_gain_db;
      return;
    }
    if (c->auto_makeup_activity_reliability < AUTO_MAKEUP_RELIABILITY_MIN) {
      double cap = c->makeup_gain_db +
                   3.0 * (c->auto_makeup_activity_reliability / AUTO_MAKEUP_RELIABILITY_MIN);
      if (c->smoothed_makeup_gain > cap)
        c->smoothed_makeup_gain = makeup_coeff * c->smoothed_makeup_gain + (1.0 - makeup_coeff) * cap;
      return;
    }
    double required_gain = c->target_lufs - c->current_lufs;
    double reliability_cap = clampd(12.0 * c->auto_makeup_activity_reliability, 3.0, 12.0);
    double headroom_cap =
        clampd(12.0 - c->limiter_feedback_gain_reduction_db * 2.0, 0.0, reliability_cap);
    double clamped_gain = clampd(required_gain, 0.0, headroom_cap);
    double diff = clamped_gain - c->smoothed_makeup_gain;
    if (fabs(diff) > 0.1) {
      c->smoothed_makeup_gain =
          makeup_coeff * c->smoothed_makeup_gain + (1.0 - makeup_coeff) * clamped_gain;
    } else {
      c->smoothed_makeup_gain = clamped_gain;
    }
  }
}

/* dsp/compressor.rs:657-678 */
double afo_compressor_compute_gain_reduction(const afo_compressor *c, double detector_db) {
  double comp_factor = 1.0 - 1.0 / c->ratio;
  if (c->knee_db <= 0.0) {
    if (detector_db <= c->threshold_db) return 0.0;
    return (detector_db - c->threshold_db) * comp_factor;
  }
  double knee_half = c->knee_db / 2.0;
  double knee_start = c->threshold_db - knee_half;
  double knee_end = c->threshold_db + knee_half;
  if (detector_db <= knee_start) return 0.0;
  if (detector_db >= knee_end) return (detector_db - c->threshold_db) * comp_factor;
  double x = detector_db - knee_start;
  return comp_factor * x * x / (2.0 * c->knee_db);
}

/* dsp/compressor.rs:681-686 */
double afo_compressor_blended_detector_db(double peak_db, double rms_db) {
  double peak_lin = afo_db_to_linear(peak_db);
  double rms_lin = afo_db_to_linear(rms_db);
  double blended = DETECTOR_PEAK_WEIGHT * peak_lin + DETECTOR_RMS_WEIGHT * rms_lin;
  return afo_linear_to_db(blended, 1e-10);
}

/* dsp/compressor.rs:725-774 */
static inline float compressor_process_sample_impl(afo_compressor *c, float input,
                                                   int update_makeup_gain) {
  if (!c->enabled) {
    c->current_gain_reduction_db = 0.0;
    return input;
  }
  double input_f64 = (double)input;
  double detector_input = process_sidechain_sample(c, input_f64);
  double detector_weight = update_sidechain_band_metrics(c, input_f64, detector_input);
  double detector_abs = fabs(detector_input);
  double inst_peak_db = afo_linear_to_db(detector_abs, 1e-10);
  double peak_coeff = inst_peak_db > c->peak_envelope_db ? c->attack_coeff : c->detector_release_coeff;
  c->peak_envelope_db = peak_coeff * c->peak_envelope_db + (1.0 - peak_coeff) * inst_peak_db;

  double input_squared = detector_input * detector_input;
  c->rms_envelope_sq = c->rms_coeff * c->rms_envelope_sq + (1.0 - c->rms_coeff) * input_squared;
  double rms_db = afo_linear_to_db(sqrt(c->rms_envelope_sq), 1e-10);

  double detector_db = afo_compressor_blended_detector_db(c->peak_envelope_db, rms_db) +
                       afo_linear_to_db(detector_weight, 1e-10);

  update_adaptive_release_time_meter(c);
  double release_diff = c->target_release_ms - c->current_release_ms;
  if (fabs(release_diff) > 1.0) {
    c->current_release_ms = c->release_smoothing_coeff * c->current_release_ms +
                            (1.0 - c->release_smoothing_coeff) * c->target_release_ms;
  } else {
    c->current_release_ms = c->target_release_ms;
  }
  c->release_coeff = afo_time_constant_to_coeff(c->current_release_ms, c->sample_rate);

  double target_gr = afo_compressor_compute_gain_reduction(c, detector_db);
  smooth_gain_reduction(c, target_gr);

  if (update_makeup_gain) {
    double speech_activity = speech_activity_from_rms_db(detector_db);
    update_auto_makeup_gain(c, speech_activity, 1.0, 1);
  }
  double output_gain =
      afo_db_to_linear(-c->current_gain_reduction_db) * afo_db_to_linear(c->smoothed_makeup_gain);
  return (float)(input_f64 * output_gain);
}

/* dsp/compressor.rs:690-692 */
float afo_compressor_process_sample(afo_compressor *c, float input) {
  return compressor_process_sample_impl(c, input, 1);
}

/* dsp/compressor.rs:700-722 */
void afo_compressor_process_block(afo_compressor *c, float *buf, size_t n,
                                  const afo_auto_makeup_input *evidence) {
  if (!c->enabled) {
    c->current_gain_reduction_db = 0.0;
    return;
  }
  double activity, reliability;
  estimate_auto_makeup_activity(c, block_rms_db(buf, n), evidence, &activity, &reliability);
  for (size_t i = 0; i < n; ++i) buf[i] = compressor_process_sample_impl(c, buf[i], 0);
  if (activity > AUTO_MAKEUP_ACTIVE_MIN && reliability >= AUTO_MAKEUP_RELIABILITY_MIN) {
    if (c->has_meter) afo_loudness_process(&c->meter, buf, n);
  }
  update_auto_makeup_gain(c, activity, reliability, n);
}

/* dsp/compressor.rs:777-798 */
void afo_compressor_reset(afo_compressor *c) {
  c->peak_envelope_db = -120.0;
  c->rms_envelope_sq = 0.0;
  c->current_gain_reduction_db = 0.0;
  c->fast_release_env_db = 0.0;
  c->slow_release_env_db = 0.0;
  c->current_release_ms = c->base_release_ms;
  c->target_release_ms = c->base_release_ms;
  c->release_coeff = afo_time_constant_to_coeff(c->current_release_ms, c->sample_rate);
  reset_sidechain_highpass_state(c);
  c->limiter_feedback_gain_reduction_db = 0.0;
  c->speech_activity_score = 0.0;
  c->auto_makeup_activity_reliability = 0.0;
  if (c->has_meter) afo_loudness_reset(&c->meter);
  c->current_lufs = -100.0;
}

/* --------------------------------------------------------------- limiter */
/* dsp/limiter.rs:29-68 (FixedMonoQueue) */
static void q_clear(afo_limiter *l) { l->q_head = 0; l->q_len = 0; }
static void q_pop_front(afo_limiter *l) {
  if (l->q_len > 0) {
    l->q_head = (l->q_head + 1) % AFO_MAX_LOOKAHEAD;
    l->q_len -= 1;
    if (l->q_len == 0) l->q_head = 0;
  }
}
static void q_pop_back(afo_limiter *l) {
  if (l->q_len > 0) {
    l->q_len -= 1;
    if (l->q_len == 0) l->q_head = 0;
  }
}
static void q_push_back(afo_limiter *l, uint64_t index, double value) {
  if (l->q_len == AFO_MAX_LOOKAHEAD) q_pop_front(l);
  size_t idx = (l->q_head + l->q_len) % AFO_MAX_LOOKAHEAD;
  l->q_index[idx] = index;
  l->q_value[idx] = value;
  l->q_len += 1;
}

static size_t lookahead_samples_for(double lookahead_ms, double sample_rate) {
  double s = round(clampd(lookahead_ms, 0.1, 10.0) / 1000.0 * sample_rate);
  size_t v = s <= 0.0 ? 0 : (size_t)s;
  return clampz(v, 1, AFO_MAX_LOOKAHEAD);
}

/* dsp/limiter.rs:106-131 */
void afo_limiter_init(afo_limiter *l, double ceiling_db, double release_ms, double sample_rate,
                      double lookahead_ms) {
  memset(l, 0, sizeof(*l));
  l->ceiling_db = ceiling_db;
  l->ceiling_linear = afo_db_to_linear(ceiling_db);
  l->release_coeff = afo_time_constant_to_coeff(release_ms, sample_rate);
  l->gain_reduction = 1.0;
  l->peak_gain_reduction_db = 0.0;
  l->sample_rate = sample_rate;
  l->lookahead_samples = lookahead_samples_for(lookahead_ms, sample_rate);
  l->enabled = 1;
}
/* dsp/limiter.rs:139-142 */
void afo_limiter_set_ceiling(afo_limiter *l, double ceiling_db) {
  l->ceiling_db = fmin(ceiling_db, 0.0);
  l->ceiling_linear = afo_db_to_linear(l->ceiling_db);
}
/* dsp/limiter.rs:150-152 */
void afo_limiter_set_release_time(afo_limiter *l, double ms) {
  l->release_coeff = afo_time_constant_to_coeff(ms, l->sample_rate);
}
/* dsp/limiter.rs:298-305 */
void afo_limiter_reset(afo_limiter *l) {
  l->gain_reduction = 1.0;
  l->peak_gain_reduction_db = 0.0;
  l->next_input_index = 0;
  l->write_idx = 0;
  memset(l->delay, 0, sizeof(float) * l->lookahead_samples);
  q_clear(l);
}
/* dsp/limiter.rs:157-166 */
void afo_limiter_set_lookahead_ms(afo_limiter *l, double ms) {
  size_t samples = lookahead_samples_for(ms, l->sample_rate);
  if (samples != l->lookahead_samples) {
    size_t old = l->lookahead_samples;
    l->lookahead_samples = samples;
    for (size_t i = old; i < samples; ++i) l->delay[i] = 0.0f; /* Vec::resize(samples, 0.0) */
    afo_limiter_reset(l);
  }
}
/* dsp/limiter.rs:179-184 */
void afo_limiter_set_enabled(afo_limiter *l, int enabled) {
  if ((l->enabled != 0) != (enabled != 0)) afo_limiter_reset(l);
  l->enabled = enabled;
}
/* dsp/limiter.rs:201-205 */
double afo_limiter_peak_gain_reduction_and_reset(afo_limiter *l) {
  double p = l->peak_gain_reduction_db;
  l->peak_gain_reduction_db = 0.0;
  return p;
}

/* dsp/limiter.rs:216-237 */
static inline void push_lookahead_sample(afo_limiter *l, double sample_abs) {
  uint64_t sample_index = l->next_input_index;
  while (l->q_len > 0) {
    size_t idx = (l->q_head + l->q_len - 1) % AFO_MAX_LOOKAHEAD;
    if (l->q_value[idx] > sample_abs) break;
    q_pop_back(l);
  }
  q_push_back(l, sample_index, sample_abs);
  if (l->next_input_index != UINT64_MAX) l->next_input_index += 1;
  uint64_t la = (uint64_t)l->lookahead_samples;
  uint64_t oldest_kept = l->next_input_index >= la ? l->next_input_index - la : 0;
  while (l->q_len > 0) {
    if (l->q_index[l->q_head] >= oldest_kept) break;
    q_pop_front(l);
  }
}

/* dsp/limiter.rs:246-284 */
float afo_limiter_process_sample(afo_limiter *l, float input) {
  if (!l->enabled) return input;
  double delayed = (double)l->delay[l->write_idx];
  double front = l->q_len > 0 ? l->q_value[l->q_head] : 0.0;
  double in_abs = fabs((double)input);
  double peak = fmax(front, in_abs);
  l->delay[l->write_idx] = input;
  push_lookahead_sample(l, in_abs);
  l->write_idx = (l->write_idx + 1) % l->lookahead_samples;
  double target_gain = peak > l->ceiling_linear ? l->ceiling_linear / peak : 1.0;
  if (target_gain < l->gain_reduction) {
    l->gain_reduction = target_gain;
  } else {
    l->gain_reduction = l->release_coeff * l->gain_reduction + (1.0 - l->release_coeff) * target_gain;
  }
  double reduction_db = l->gain_reduction < 1.0 ? -afo_linear_to_db(l->gain_reduction, 1e-10) : 0.0;
  if (reduction_db > l->peak_gain_reduction_db) l->peak_gain_reduction_db = reduction_db;
  double limited = delayed * l->gain_reduction;
  return (float)clampd(limited, -l->ceiling_linear, l->ceiling_linear);
}
/* dsp/limiter.rs:287-295 */
void afo_limiter_process_block(afo_limiter *l, float *buf, size_t n) {
  if (!l->enabled) return;
  for (size_t i = 0; i < n; ++i) buf[i] = afo_limiter_process_sample(l, buf[i]);
}

/* ------------------------------------------------------------- true peak */
/* dsp/true_peak.rs:173-186 */
float afo_tp_observe(afo_tp_oversampler *o, float sample) {
  memmove(&o->history[1], &o->history[0], sizeof(float) * (AFO_TP_TAPS - 1));
  o->history[0] = sample;
  float peak = fabsf(sample);
  for (int p = 0; p < 4; ++p) {
    float interpolated = 0.0f;
    for (int k = 0; k < AFO_TP_TAPS; ++k) interpolated = fmaf(AFO_TP_FIR[p][k], o->history[k], interpolated);
    peak = fmaxf(peak, fabsf(interpolated));
  }
  return peak;
}
void afo_tp_detector_init(afo_tp_detector *d) { memset(d, 0, sizeof(*d)); }
/* dsp/true_peak.rs:208-218 */
float afo_tp_detector_process_block(afo_tp_detector *d, const float *samples, size_t n) {
  float peak = 0.0f;
  for (size_t i = 0; i < n; ++i) {
    float s = isfinite(samples[i]) ? samples[i] : 0.0f;
    peak = fmaxf(peak, afo_tp_observe(&d->os, s));
  }
  d->last_peak = peak;
  return peak;
}
/* dsp/true_peak.rs:308-313 */
void afo_tp_limiter_set_release_ms(afo_tp_limiter *l, float release_ms) {
  l->release_coeff = (float)afo_time_constant_to_coeff((double)clampf(release_ms, 5.0f, 500.0f),
                                                       (double)l->sample_rate);
}
/* dsp/true_peak.rs:266-283 */
void afo_tp_limiter_init(afo_tp_limiter *l, float sample_rate, float ceiling_db, float release_ms) {
  memset(l, 0, sizeof(*l));
  l->ceiling_linear = (float)afo_db_to_linear((double)ceiling_db);
  l->release_coeff = (float)afo_time_constant_to_coeff((double)release_ms, (double)sample_rate);
  l->gain_reduction = 1.0f;
  l->sample_rate = fmaxf(sample_rate, 1.0f);
  afo_tp_limiter_set_release_ms(l, release_ms);
}
/* dsp/true_peak.rs:289-298 */
void afo_tp_limiter_reset(afo_tp_limiter *l) {
  l->gain_reduction = 1.0f;
  memset(l->delay, 0, sizeof(l->delay));
  l->write_idx = 0;
  memset(&l->in_os, 0, sizeof(l->in_os));
  memset(&l->out_os, 0, sizeof(l->out_os));
  l->last_input_true_peak = 0.0f;
  l->last_output_true_peak = 0.0f;
  l->peak_gain_reduction_db = 0.0f;
}
/* dsp/true_peak.rs:304-306 */
void afo_tp_limiter_set_ceiling_linear(afo_tp_limiter *l, float ceiling_linear) {
  l->ceiling_linear = clampf(ceiling_linear, 0.000001f, 1.0f);
}
/* dsp/true_peak.rs:315-321 */
static inline float tp_current_gain_reduction_db(const afo_tp_limiter *l) {
  if (l->gain_reduction >= 1.0f) return 0.0f;
  return -20.0f * log10f(fmaxf(l->gain_reduction, 1e-10f));
}
/* dsp/true_peak.rs:337-378 */
afo_tp_block_stats afo_tp_limiter_process_block(afo_tp_limiter *l, float *samples, size_t n) {
  afo_tp_block_stats stats = {0, 0.0f, 0.0f, 0.0f};
  int limited = 0;
  for (size_t i = 0; i < n; ++i) {
    float input = isfinite(samples[i]) ? samples[i] : 0.0f;
    float delayed = l->delay[l->write_idx];
    l->delay[l->write_idx] = input;
    l->write_idx = (l->write_idx + 1) % AFO_TP_LOOKAHEAD;

    float input_true_peak = afo_tp_observe(&l->in_os, input);
    l->last_input_true_peak = input_true_peak;
    stats.input_true_peak = fmaxf(stats.input_true_peak, input_true_peak);

    float target_gain = 1.0f;
    if (input_true_peak > l->ceiling_linear)
      target_gain = clampf((l->ceiling_linear * 0.999f) / input_true_peak, 0.0f, 1.0f);
    if (target_gain < l->gain_reduction) {
      l->gain_reduction = target_gain;
      limited = 1;
    } else {
      l->gain_reduction = l->release_coeff * l->gain_reduction + (1.0f - l->release_coeff) * target_gain;
    }
    float reduction_db = tp_current_gain_reduction_db(l);
    l->peak_gain_reduction_db = fmaxf(l->peak_gain_reduction_db, reduction_db);
    stats.max_gain_reduction_db = fmaxf(stats.max_gain_reduction_db, reduction_db);

    float output = clampf(delayed * l->gain_reduction, -l->ceiling_linear, l->ceiling_linear);
    output = isfinite(output) ? output : 0.0f;
    float out_tp = afo_tp_observe(&l->out_os, output);
    l->last_output_true_peak = out_tp;
    stats.output_true_peak = fmaxf(stats.output_true_peak, out_tp);
    samples[i] = output;
  }
  stats.limited_events = (uint64_t)limited;
  return stats;
}

/* -------------------------------------------------------------- de-esser */
#define VOICE_REFERENCE_SIDECHAIN_DISCOUNT 0.6
#define DETECTOR_RATIO_GATE_DB 1.5
#define DETECTOR_RATIO_FULL_DB 10.0
#define DETECTOR_LEVEL_GATE_DB (-62.0)
#define DETECTOR_LEVEL_FULL_DB (-24.0)
#define DETECTOR_VOICE_GATE_DB (-58.0)
#define DETECTOR_VOICE_FULL_DB (-34.0)
#define AUTO_BASELINE_FALL_MS 13.88
#define AUTO_BASELINE_RISE_MS 34.72
#define AUTO_BASELINE_INACTIVE_DECAY_MS 20.82
#define DEESSER_DEFAULT_HIGH_CUT_HZ 11000.0
#define BROADBAND_NARROWNESS_GATE 0.34
#define BROADBAND_NARROWNESS_FULL 0.68

/* dsp/deesser.rs:263-272 */
static double dynamic_eq_center_hz(double lo, double hi) { return sqrt(lo * hi); }
static double dynamic_eq_q(double lo, double hi) {
  double bandwidth = fmax(hi - lo, 200.0);
  return clampd(dynamic_eq_center_hz(lo, hi) / bandwidth, 0.5, 6.0);
}
/* dsp/deesser.rs:47-63 */
static void deesser_band_init(afo_deesser_band *b, double lo, double hi, double fs) {
  memset(b, 0, sizeof(*b));
  b->low_hz = lo;
  b->high_hz = hi;
  afo_biquad_init(&b->detector_hp, AFO_BQ_HIGHPASS, lo, 0.0, 0.707, fs);
  afo_biquad_init(&b->detector_lp, AFO_BQ_LOWPASS, hi, 0.0, 0.707, fs);
  afo_biquad_init(&b->dynamic_eq, AFO_BQ_PEAKING, dynamic_eq_center_hz(lo, hi), 0.0,
                  dynamic_eq_q(lo, hi), fs);
}
/* dsp/deesser.rs:65-74 */
static void deesser_band_set_bounds(afo_deesser_band *b, double lo, double hi) {
  b->low_hz = lo;
  b->high_hz = hi;
  afo_biquad_set_frequency(&b->detector_hp, lo);
  afo_biquad_set_frequency(&b->detector_lp, hi);
  afo_biquad_set_frequency(&b->dynamic_eq, dynamic_eq_center_hz(lo, hi));
  afo_biquad_set_q(&b->dynamic_eq, dynamic_eq_q(lo, hi));
}
/* dsp/deesser.rs:76-85 */
static void deesser_band_reset(afo_deesser_band *b) {
  b->env = 0.0;
  b->confidence = 0.0;
  b->baseline_excess_db = 0.0;
  b->reduction_db = 0.0;
  afo_biquad_reset(&b->detector_hp);
  afo_biquad_reset(&b->detector_lp);
  afo_biquad_reset(&b->dynamic_eq);
  afo_biquad_set_gain_db_immediate(&b->dynamic_eq, 0.0);
}
/* dsp/deesser.rs:110-136, 247-261 */
void afo_deesser_init(afo_deesser *d, double fs) {
  memset(d, 0, sizeof(*d));
  double lo = 4000.0, hi = DEESSER_DEFAULT_HIGH_CUT_HZ;
  double span = fmax(hi - lo, 600.0);
  double split_a = lo + span / 3.0;
  double split_b = lo + span * 2.0 / 3.0;
  deesser_band_init(&d->bands[0], lo, split_a, fs);
  deesser_band_init(&d->bands[1], split_a, split_b, fs);
  deesser_band_init(&d->bands[2], split_b, hi, fs);
  d->enabled = 0;
  d->auto_enabled = 1;
  d->auto_amount = 0.5;
  d->threshold_db = -28.0;
  d->ratio = 4.0;
  d->attack_coeff = afo_time_constant_to_coeff(2.0, fs);
  d->release_coeff = afo_time_constant_to_coeff(80.0, fs);
  d->detector_attack_coeff = afo_time_constant_to_coeff(1.5, fs);
  d->detector_release_coeff = afo_time_constant_to_coeff(60.0, fs);
  d->max_reduction_db = 6.0;
  d->low_cut_hz = lo;
  d->high_cut_hz = hi;
  d->sample_rate = fs;
}
/* dsp/deesser.rs:231-245 */
static void deesser_rebuild_detector_filters(afo_deesser *d) {
  double span = fmax(d->high_cut_hz - d->low_cut_hz, 600.0);
  double split_a = d->low_cut_hz + span / 3.0;
  double split_b = d->low_cut_hz + span * 2.0 / 3.0;
  deesser_band_set_bounds(&d->bands[0], d->low_cut_hz, split_a);
  deesser_band_set_bounds(&d->bands[1], split_a, split_b);
  deesser_band_set_bounds(&d->bands[2], split_b, d->high_cut_hz);
}
void afo_deesser_set_enabled(afo_deesser *d, int e) { d->enabled = e; }
void afo_deesser_set_auto_enabled(afo_deesser *d, int e) { d->auto_enabled = e; }
void afo_deesser_set_auto_amount(afo_deesser *d, double v) { d->auto_amount = clampd(v, 0.0, 1.0); }
/* dsp/deesser.rs:319-325 */
void afo_deesser_set_low_cut_hz(afo_deesser *d, double v) {
  d->low_cut_hz = clampd(v, 2000.0, 12000.0);
  if (d->high_cut_hz <= d->low_cut_hz + 200.0)
    d->high_cut_hz = clampd(d->low_cut_hz + 200.0, 2200.0, 16000.0);
  deesser_rebuild_detector_filters(d);
}
/* dsp/deesser.rs:328-334 */
void afo_deesser_set_high_cut_hz(afo_deesser *d, double v) {
  d->high_cut_hz = clampd(v, 2200.0, 16000.0);
  if (d->high_cut_hz <= d->low_cut_hz + 200.0)
    d->low_cut_hz = clampd(d->high_cut_hz - 200.0, 2000.0, 12000.0);
  deesser_rebuild_detector_filters(d);
}
void afo_deesser_set_threshold_db(afo_deesser *d, double v) { d->threshold_db = clampd(v, -60.0, -6.0); }
void afo_deesser_set_ratio(afo_deesser *d, double v) { d->ratio = clampd(v, 1.0, 20.0); }
void afo_deesser_set_attack_ms(afo_deesser *d, double v) {
  d->attack_coeff = afo_time_constant_to_coeff(clampd(v, 0.1, 50.0), d->sample_rate);
}
void afo_deesser_set_release_ms(afo_deesser *d, double v) {
  d->release_coeff = afo_time_constant_to_coeff(clampd(v, 5.0, 500.0), d->sample_rate);
}
void afo_deesser_set_max_reduction_db(afo_deesser *d, double v) {
  d->max_reduction_db = clampd(v, 0.0, 24.0);
}

/* dsp/deesser.rs:149-157 */
static inline double smooth_value(double prev, double input, double attack, double release) {
  double coeff = input > prev ? attack : release;
  return coeff * prev + (1.0 - coeff) * input;
}
static inline double lerp(double a, double b, double t) { return a + (b - a) * t; }
static inline double normalize_range(double value, double start, double end) {
  return clampd((value - start) / (end - start), 0.0, 1.0);
}
/* dsp/deesser.rs:169-171 */
static inline double confidence_reduction_gain(double confidence, double floor_) {
  return normalize_range(confidence, clampd(floor_, 0.0, 0.95), 1.0);
}
/* dsp/deesser.rs:173-224 */
static double detector_confidence_target(double sidechain_level_db, double voice_reference_db,
                                         double narrowness) {
  double spectral_ratio_db = fmax(sidechain_level_db - voice_reference_db, 0.0);
  double ratio_conf = normalize_range(spectral_ratio_db, DETECTOR_RATIO_GATE_DB, DETECTOR_RATIO_FULL_DB);
  double level_conf = normalize_range(sidechain_level_db, DETECTOR_LEVEL_GATE_DB, DETECTOR_LEVEL_FULL_DB);
  double voice_conf = normalize_range(voice_reference_db, DETECTOR_VOICE_GATE_DB, DETECTOR_VOICE_FULL_DB);
  double narrow_support = (spectral_ratio_db > 6.0 && sidechain_level_db > -45.0) ? 0.75 : 0.0;
  double voice_support = fmax(voice_conf, narrow_support);
  double balance_conf = ratio_conf > 0.12 ? fmax(ratio_conf, voice_support * 0.65) : ratio_conf;
  double broadband_penalty = lerp(0.35, 1.0, balance_conf);
  double narrowness_gain = lerp(
      0.35, 1.0, normalize_range(narrowness, BROADBAND_NARROWNESS_GATE, BROADBAND_NARROWNESS_FULL));
  return (0.62 * ratio_conf + 0.18 * level_conf + 0.20 * voice_support) * broadband_penalty *
         narrowness_gain;
}

/* dsp/deesser.rs:405-547 */
float afo_deesser_process_sample(afo_deesser *d, float input) {
  if (!d->enabled) {
    d->current_reduction_db = 0.0;
    d->detector_confidence = 0.0;
    return input;
  }
  double broadband_level = (double)fabsf(input);
  d->broadband_env = smooth_value(d->broadband_env, broadband_level, d->detector_attack_coeff,
                                  d->detector_release_coeff);
  double detector_attack = d->detector_attack_coeff;
  double detector_release = d->detector_release_coeff;
  double band_level_db[3] = {0.0, 0.0, 0.0};
  double total_sibilance_env = 0.0, max_sibilance_env = 0.0;
  for (int i = 0; i < 3; ++i) {
    afo_deesser_band *b = &d->bands[i];
    float sc_hp = afo_biquad_process_sample(&b->detector_hp, input);
    float sc = afo_biquad_process_sample(&b->detector_lp, sc_hp);
    b->env = smooth_value(b->env, (double)fabsf(sc), detector_attack, detector_release);
    total_sibilance_env += b->env;
    max_sibilance_env = fmax(max_sibilance_env, b->env);
    band_level_db[i] = afo_linear_to_db(b->env, 1e-10);
  }
  double voice_reference_level =
      fmax(d->broadband_env - total_sibilance_env * VOICE_REFERENCE_SIDECHAIN_DISCOUNT, 1e-8);
  double voice_reference_db = afo_linear_to_db(voice_reference_level, 1e-10);
  double narrowness = total_sibilance_env > 1e-10 ? max_sibilance_env / total_sibilance_env : 0.0;

  double amount = clampd(d->auto_amount, 0.0, 1.0);
  double trigger_offset_db = lerp(8.0, 0.8, amount);
  double slope = lerp(0.08, 1.9, amount);
  double auto_cap = lerp(0.8, 14.0, amount);
  double confidence_floor = lerp(0.28, 0.06, amount);
  double baseline_fall = afo_time_constant_to_coeff(AUTO_BASELINE_FALL_MS, d->sample_rate);
  double baseline_rise = afo_time_constant_to_coeff(AUTO_BASELINE_RISE_MS, d->sample_rate);
  double baseline_inactive = afo_time_constant_to_coeff(AUTO_BASELINE_INACTIVE_DECAY_MS, d->sample_rate);
  double target_reductions[3] = {0.0, 0.0, 0.0};
  double target_sum = 0.0, aggregate_confidence = 0.0;

  for (int i = 0; i < 3; ++i) {
    double sidechain_level_db = band_level_db[i];
    double spectral_ratio_db = fmax(sidechain_level_db - voice_reference_db, 0.0);
    double band_dominance = max_sibilance_env > 1e-10 ? sqrt(d->bands[i].env / max_sibilance_env) : 0.0;
    double confidence_target =
        detector_confidence_target(sidechain_level_db, voice_reference_db, narrowness) * band_dominance;
    afo_deesser_band *b = &d->bands[i];
    b->confidence = smooth_value(b->confidence, clampd(confidence_target, 0.0, 1.0), detector_attack,
                                 detector_release);
    aggregate_confidence = fmax(aggregate_confidence, b->confidence);

    double target_reduction;
    if (d->auto_enabled) {
      int voice_active = voice_reference_db > -55.0 || sidechain_level_db > -55.0;
      if (voice_active) {
        double baseline_target = clampd(spectral_ratio_db * 0.45, 0.0, 24.0);
        double baseline_coeff = baseline_target < b->baseline_excess_db ? baseline_fall : baseline_rise;
        b->baseline_excess_db =
            baseline_coeff * b->baseline_excess_db + (1.0 - baseline_coeff) * baseline_target;
      } else {
        b->baseline_excess_db *= baseline_inactive;
      }
      double cap_db = fmin(auto_cap, d->max_reduction_db * 0.75);
      double confidence_gain = confidence_reduction_gain(b->confidence, confidence_floor);
      double over_db = fmax(spectral_ratio_db - b->baseline_excess_db - trigger_offset_db, 0.0);
      target_reduction = clampd(over_db * slope * confidence_gain, 0.0, cap_db);
    } else if (sidechain_level_db > d->threshold_db) {
      double ratio_threshold_db = clampd((d->threshold_db + 60.0) * 0.10, 0.0, 6.0);
      double level_over_db = sidechain_level_db - d->threshold_db;
      double ratio_over_db = spectral_ratio_db - ratio_threshold_db;
      if (ratio_over_db > 0.0) {
        double over_db = fmin(level_over_db, ratio_over_db);
        double confidence_gain = confidence_reduction_gain(b->confidence, 0.22);
        target_reduction = clampd((1.0 - (1.0 / d->ratio)) * over_db * confidence_gain, 0.0,
                                  d->max_reduction_db * 0.75);
      } else {
        target_reduction = 0.0;
      }
    } else {
      target_reduction = 0.0;
    }
    target_reductions[i] = target_reduction;
    target_sum += target_reduction;
  }

  if (target_sum > d->max_reduction_db && target_sum > 0.0) {
    double scale = d->max_reduction_db / target_sum;
    for (int i = 0; i < 3; ++i) target_reductions[i] *= scale;
  }

  float processed = input;
  double total_reduction = 0.0;
  for (int i = 0; i < 3; ++i) {
    afo_deesser_band *b = &d->bands[i];
    b->reduction_db = smooth_value(b->reduction_db, target_reductions[i], d->attack_coeff, d->release_coeff);
    total_reduction += b->reduction_db;
    double dynamic_gain_db = -b->reduction_db;
    if (fabs(b->dynamic_eq.gain_db - dynamic_gain_db) > 0.001)
      afo_biquad_set_gain_db_immediate(&b->dynamic_eq, dynamic_gain_db);
    processed = afo_biquad_process_sample(&b->dynamic_eq, processed);
  }
  d->current_reduction_db = fmin(total_reduction, d->max_reduction_db);
  d->detector_confidence = clampd(aggregate_confidence, 0.0, 1.0);
  return processed;
}
/* dsp/deesser.rs:550-560 */
void afo_deesser_process_block(afo_deesser *d, float *buf, size_t n) {
  if (!d->enabled) {
    d->current_reduction_db = 0.0;
    d->detector_confidence = 0.0;
    return;
  }
  for (size_t i = 0; i < n; ++i) buf[i] = afo_deesser_process_sample(d, buf[i]);
}
/* dsp/deesser.rs:563-570 */
void afo_deesser_reset(afo_deesser *d) {
  d->current_reduction_db = 0.0;
  d->broadband_env = 0.0;
  d->detector_confidence = 0.0;
  for (int i = 0; i < 3; ++i) deesser_band_reset(&d->bands[i]);
}

/* ------------------------------------------------------------- prefilter */
/* audio/processor.rs:74-76 */
void afo_prefilter_init(afo_prefilter *p, double sample_rate) {
  p->dc_x1 = 0.0f;
  p->dc_y1 = 0.0f;
  afo_biquad_init(&p->hp, AFO_BQ_HIGHPASS, 80.0, 0.0, 0.707, sample_rate);
}
/* audio/processor/routing.rs:826-843 */
void afo_prefilter_process_block(afo_prefilter *p, float *buf, size_t n, int apply_fixed_highpass) {
  const float coeff = 0.995f;
  for (size_t i = 0; i < n; ++i) {
    float input = buf[i];
    float output = input - p->dc_x1 + coeff * p->dc_y1;
    p->dc_x1 = input;
    p->dc_y1 = output;
    buf[i] = apply_fixed_highpass ? afo_biquad_process_sample(&p->hp, output) : output;
  }
}
/* audio/processor/routing.rs:802-823 */
uint64_t afo_sanitize_and_clamp(float *buf, size_t n) {
  uint64_t clipped = 0;
  for (size_t i = 0; i < n; ++i) {
    if (!isfinite(buf[i])) {
      buf[i] = 0.0f;
      continue;
    }
    if (fabsf(buf[i]) > 1.0f) clipped += 1;
    buf[i] = clampf(buf[i], -1.0f, 1.0f);
  }
  return clipped;
}

/* ------------------------------------------------- offline block processor */
/* audio/processor/block_processor.rs:46-60 */
afo_chain *afo_chain_new(double sample_rate) {
  afo_chain *c = (afo_chain *)calloc(1, sizeof(afo_chain));
  if (!c) return NULL;
  afo_deesser_init(&c->deesser, sample_rate);
  afo_eq_init(&c->eq, sample_rate);
  afo_compressor_init(&c->compressor, -18.0, 3.0, 5.0, 100.0, 0.0, 6.0, sample_rate);
  afo_limiter_init(&c->limiter, -0.5, 50.0, sample_rate, 2.0);
  afo_tp_limiter_init(&c->tp_limiter, (float)sample_rate, -1.5f, 80.0f);
  afo_tp_detector_init(&c->tp_detector);
  c->deesser_enabled = 0;
  c->eq_enabled = 1;
  c->compressor_enabled = 0;
  c->limiter_enabled = 1;
  c->eq_before_deesser = 0;
  return c;
}
void afo_chain_free(afo_chain *c) {
  if (!c) return;
  afo_compressor_free(&c->compressor);
  free(c);
}
/* audio/processor/block_processor.rs:62-84 */
void afo_chain_set_deesser_enabled(afo_chain *c, int e) { c->deesser_enabled = e; afo_deesser_set_enabled(&c->deesser, e); }
void afo_chain_set_eq_enabled(afo_chain *c, int e) { c->eq_enabled = e; c->eq.enabled = e; }
void afo_chain_set_compressor_enabled(afo_chain *c, int e) { c->compressor_enabled = e; afo_compressor_set_enabled(&c->compressor, e); }
void afo_chain_set_limiter_enabled(afo_chain *c, int e) { c->limiter_enabled = e; afo_limiter_set_enabled(&c->limiter, e); }
void afo_chain_set_eq_before_deesser(afo_chain *c, int e) { c->eq_before_deesser = e; }
afo_deesser *afo_chain_deesser(afo_chain *c) { return &c->deesser; }
afo_eq *afo_chain_eq(afo_chain *c) { return &c->eq; }
afo_compressor *afo_chain_compressor(afo_chain *c) { return &c->compressor; }
afo_limiter *afo_chain_limiter(afo_chain *c) { return &c->limiter; }
afo_tp_limiter *afo_chain_tp_limiter(afo_chain *c) { return &c->tp_limiter; }

static float block_abs_peak(const float *buf, size_t n) {
  float peak = 0.0f;
  for (size_t i = 0; i < n; ++i) peak = fmaxf(peak, fabsf(buf[i]));
  return peak;
}

/* audio/processor/block_processor.rs:106-161 (the copy into `output` is done by the caller) */
afo_block_stats afo_chain_process_block(afo_chain *c, float *block, size_t n) {
  afo_block_stats stats;
  memset(&stats, 0, sizeof(stats));
  stats.input_sample_peak = block_abs_peak(block, n);
  if (c->eq_before_deesser) {
    if (c->eq_enabled) afo_eq_process_block(&c->eq, block, n);
    if (c->deesser_enabled) {
      afo_deesser_process_block(&c->deesser, block, n);
      stats.deesser_gain_reduction_db = (float)c->deesser.current_reduction_db;
    }
  } else {
    if (c->deesser_enabled) {
      afo_deesser_process_block(&c->deesser, block, n);
      stats.deesser_gain_reduction_db = (float)c->deesser.current_reduction_db;
    }
    if (c->eq_enabled) afo_eq_process_block(&c->eq, block, n);
  }
  if (c->compressor_enabled) {
    afo_compressor_process_block(&c->compressor, block, n, NULL);
    stats.compressor_gain_reduction_db = (float)c->compressor.current_gain_reduction_db;
  }
  if (c->limiter_enabled) {
    afo_limiter_process_block(&c->limiter, block, n);
    stats.limiter_peak_gain_reduction_db = (float)afo_limiter_peak_gain_reduction_and_reset(&c->limiter);
    afo_tp_limiter_set_ceiling_linear(&c->tp_limiter, powf(10.0f, (float)c->limiter.ceiling_db / 20.0f));
    afo_tp_block_stats tp = afo_tp_limiter_process_block(&c->tp_limiter, block, n);
    stats.true_peak_limiter_input_peak = tp.input_true_peak;
    stats.true_peak_limiter_gain_reduction_db = tp.max_gain_reduction_db;
    stats.true_peak_limited_events = tp.limited_events;
  }
  stats.output_sample_peak = block_abs_peak(block, n);
  stats.output_true_peak = afo_tp_detector_process_block(&c->tp_detector, block, n);
  return stats;
}

/* ------------------------------------------------ simulate_auto_eq_chain */
/* audio/processor/python_api.rs:54-56 */
float afo_linear_to_db_f32(float v) { return 20.0f * log10f(fmaxf(v, 1.0e-12f)); }

static int cmp_f32_total(const void *a, const void *b) {
  /* f32::total_cmp */
  int32_t x, y;
  memcpy(&x, a, 4);
  memcpy(&y, b, 4);
  x ^= (int32_t)(((uint32_t)(x >> 31)) >> 1);
  y ^= (int32_t)(((uint32_t)(y >> 31)) >> 1);
  return (x > y) - (x < y);
}
/* audio/processor/python_api.rs:58-72 (sorts `values` in place) */
float afo_percentile_f32(float *values, size_t n, float percentile) {
  if (n == 0) return 0.0f;
  qsort(values, n, sizeof(float), cmp_f32_total);
  float position = (float)(n - 1) * clampf(percentile, 0.0f, 1.0f);
  size_t lower = (size_t)floorf(position);
  size_t upper = (size_t)ceilf(position);
  if (lower == upper) return values[lower];
  float fraction = position - (float)lower;
  return values[lower] + fraction * (values[upper] - values[lower]);
}

/* audio/processor/python_api.rs:74-111 */
float afo_pumping_score(const float *gr, size_t n, float cadence_hz) {
  if (n < 3 || !isfinite(cadence_hz) || cadence_hz <= 0.0f) return 0.0f;
  const float pi = 3.14159265358979323846f;
  float dt = 1.0f / cadence_hz;
  float highpass_rc = 1.0f / (2.0f * pi * 2.0f);
  float lowpass_rc = 1.0f / (2.0f * pi * 8.0f);
  float highpass_alpha = highpass_rc / (highpass_rc + dt);
  float lowpass_alpha = dt / (lowpass_rc + dt);
  float previous_input = gr[0];
  float highpass = 0.0f, bandpass = 0.0f;
  size_t m = n - 1;
  float *bandpass_abs = (float *)malloc(sizeof(float) * m);
  float *sorted = (float *)malloc(sizeof(float) * m);
  float *deltas = (float *)malloc(sizeof(float) * m);
  for (size_t i = 1; i < n; ++i) {
    float value = gr[i];
    if (!isfinite(value)) {
      free(bandpass_abs); free(sorted); free(deltas);
      return INFINITY;
    }
    highpass = highpass_alpha * (highpass + value - previous_input);
    bandpass += lowpass_alpha * (highpass - bandpass);
    bandpass_abs[i - 1] = fabsf(bandpass);
    deltas[i - 1] = fabsf(value - previous_input);
    previous_input = value;
  }
  memcpy(sorted, bandpass_abs, sizeof(float) * m);
  float robust_limit = afo_percentile_f32(sorted, m, 0.95f);
  float sum = 0.0f;
  for (size_t i = 0; i < m; ++i) {
    float v = fminf(bandpass_abs[i], robust_limit);
    sum += v * v;
  }
  float robust_rms = m == 0 ? 0.0f : sqrtf(sum / (float)m);
  float p95 = afo_percentile_f32(deltas, m, 0.95f);
  free(bandpass_abs); free(sorted); free(deltas);
  return robust_rms + p95;
}

/* audio/processor/python_api.rs:415-487 */
void afo_sim_settings_default(afo_sim_settings *s) {
  memset(s, 0, sizeof(*s));
  s->deesser_enabled = 0;
  s->deesser_auto_enabled = 1;
  s->deesser_auto_amount = 0.5;
  s->deesser_low_cut_hz = 4000.0;
  s->deesser_high_cut_hz = 11000.0;
  s->deesser_threshold_db = -28.0;
  s->deesser_ratio = 4.0;
  s->deesser_attack_ms = 2.0;
  s->deesser_release_ms = 80.0;
  s->deesser_max_reduction_db = 6.0;
  s->eq_before_deesser = 0;
  s->compressor_enabled = 1;
  s->compressor_threshold_db = -20.0;
  s->compressor_ratio = 4.0;
  s->compressor_attack_ms = 10.0;
  s->compressor_release_ms = 200.0;
  s->compressor_makeup_gain_db = 0.0;
  s->compressor_adaptive_release = 0;
  s->compressor_base_release_ms = 50.0;
  s->compressor_auto_makeup_enabled = 0;
  s->compressor_target_lufs = -18.0;
  s->compressor_sidechain_highpass_enabled = 1;
  s->limiter_enabled = 1;
  s->limiter_ceiling_db = -0.5;
  s->limiter_careful_output_enabled = 1;
  s->limiter_lookahead_ms = 2.0;
  s->limiter_release_ms = 50.0;
}

typedef struct { float in_db, out_db, comp_gr, deesser_gr; } analysis_row;

/* audio/processor/python_api.rs:378-714 */
int afo_simulate_auto_eq_chain(const float *audio, size_t n, double sample_rate,
                               const double bands[AFO_NUM_BANDS][3],
                               const afo_sim_settings *s, afo_sim_result *r, float *out_audio) {
  afo_sim_settings defaults;
  if (!s) {
    afo_sim_settings_default(&defaults);
    s = &defaults;
  }
  if (!isfinite(sample_rate) || sample_rate <= 0.0) return -1;
  afo_chain *p = afo_chain_new(sample_rate);
  afo_chain_set_eq_enabled(p, 1);
  if (s->has_eq_bands_v2) {
    for (size_t i = 0; i < AFO_NUM_BANDS; ++i) afo_eq_set_band_config(&p->eq, i, &s->eq_bands_v2[i]);
    afo_eq_reset(&p->eq);
  } else {
    for (size_t i = 0; i < AFO_NUM_BANDS; ++i) {
      afo_eq_set_band_frequency(&p->eq, i, bands[i][0]);
      afo_eq_set_band_gain(&p->eq, i, bands[i][1]);
      afo_eq_set_band_q(&p->eq, i, bands[i][2]);
    }
  }
  afo_chain_set_eq_before_deesser(p, s->eq_before_deesser);
  afo_chain_set_deesser_enabled(p, s->deesser_enabled);
  if (s->deesser_enabled) {
    afo_deesser *d = &p->deesser;
    afo_deesser_set_auto_enabled(d, s->deesser_auto_enabled);
    afo_deesser_set_auto_amount(d, s->deesser_auto_amount);
    afo_deesser_set_low_cut_hz(d, s->deesser_low_cut_hz);
    afo_deesser_set_high_cut_hz(d, s->deesser_high_cut_hz);
    afo_deesser_set_threshold_db(d, s->deesser_threshold_db);
    afo_deesser_set_ratio(d, s->deesser_ratio);
    afo_deesser_set_attack_ms(d, s->deesser_attack_ms);
    afo_deesser_set_release_ms(d, s->deesser_release_ms);
    afo_deesser_set_max_reduction_db(d, s->deesser_max_reduction_db);
  }
  afo_chain_set_compressor_enabled(p, s->compressor_enabled);
  if (s->compressor_enabled) {
    afo_compressor *c = &p->compressor;
    afo_compressor_set_threshold(c, s->compressor_threshold_db);
    afo_compressor_set_ratio(c, s->compressor_ratio);
    afo_compressor_set_attack_time(c, s->compressor_attack_ms);
    afo_compressor_set_release_time(c, s->compressor_release_ms);
    afo_compressor_set_makeup_gain(c, s->compressor_makeup_gain_db);
    afo_compressor_set_adaptive_release(c, s->compressor_adaptive_release);
    afo_compressor_set_base_release_time(c, s->compressor_base_release_ms);
    afo_compressor_set_auto_makeup_enabled(c, s->compressor_auto_makeup_enabled);
    afo_compressor_set_target_lufs(c, s->compressor_target_lufs);
    afo_compressor_set_sidechain_highpass_enabled(c, s->compressor_sidechain_highpass_enabled);
  }
  afo_chain_set_limiter_enabled(p, s->limiter_enabled);
  /* audio/processor/control.rs:904-910, CAREFUL_OUTPUT_CEILING_DB = -1.5 */
  double eff = s->limiter_careful_output_enabled ? fmin(s->limiter_ceiling_db, -1.5) : s->limiter_ceiling_db;
  float effective_ceiling_db = (float)eff;
  if (s->limiter_enabled) {
    afo_limiter_set_lookahead_ms(&p->limiter, s->limiter_lookahead_ms);
    afo_limiter_set_ceiling(&p->limiter, (double)effective_ceiling_db);
    afo_limiter_set_release_time(&p->limiter, s->limiter_release_ms);
    afo_tp_limiter_set_release_ms(&p->tp_limiter, (float)s->limiter_release_ms);
  }

  double input_square_sum = 0.0, output_square_sum = 0.0;
  size_t input_samples = 0, output_samples = 0;
  float input_sample_peak = 0.0f, output_sample_peak = 0.0f, pre_limiter_true_peak = 0.0f;
  float output_true_peak = 0.0f, limiter_gr = 0.0f, tp_gr = 0.0f, comp_gr = 0.0f, deesser_gr = 0.0f;
  uint64_t tp_events = 0;
  int non_finite_output = 0;

  double blk = round(sample_rate * 0.020);
  size_t block_samples = clampz(blk <= 0.0 ? 0 : (size_t)blk, 1, 8192); /* RT_PROCESS_BUFFER_CAPACITY */
  size_t n_rows = (n + block_samples - 1) / block_samples;
  analysis_row *rows = (analysis_row *)malloc(sizeof(analysis_row) * (n_rows ? n_rows : 1));
  float *block = (float *)malloc(sizeof(float) * block_samples);
  size_t row_count = 0;

  for (size_t start = 0; start < n; start += block_samples) {
    size_t len = n - start < block_samples ? n - start : block_samples;
    double block_in_sq = 0.0;
    for (size_t i = 0; i < len; ++i) {
      float v = audio[start + i];
      if (!isfinite(v)) v = 0.0f;
      block[i] = v;
      input_square_sum += (double)v * (double)v;
      block_in_sq += (double)v * (double)v;
      input_samples += 1;
    }
    afo_block_stats st = afo_chain_process_block(p, block, len);
    float block_input_rms = (float)sqrt(block_in_sq / (double)len);
    double block_out_sq = 0.0;
    for (size_t i = 0; i < len; ++i) {
      if (!isfinite(block[i])) {
        non_finite_output = 1;
      } else {
        block_out_sq += (double)block[i] * (double)block[i];
      }
    }
    float block_output_rms = (float)sqrt(block_out_sq / (double)len);
    rows[row_count].in_db = afo_linear_to_db_f32(block_input_rms);
    rows[row_count].out_db = afo_linear_to_db_f32(block_output_rms);
    rows[row_count].comp_gr = st.compressor_gain_reduction_db;
    rows[row_count].deesser_gr = st.deesser_gain_reduction_db;
    row_count += 1;
    input_sample_peak = fmaxf(input_sample_peak, st.input_sample_peak);
    output_sample_peak = fmaxf(output_sample_peak, st.output_sample_peak);
    pre_limiter_true_peak = fmaxf(pre_limiter_true_peak, st.true_peak_limiter_input_peak);
    output_true_peak = fmaxf(output_true_peak, st.output_true_peak);
    limiter_gr = fmaxf(limiter_gr, st.limiter_peak_gain_reduction_db);
    tp_gr = fmaxf(tp_gr, st.true_peak_limiter_gain_reduction_db);
    comp_gr = fmaxf(comp_gr, st.compressor_gain_reduction_db);
    deesser_gr = fmaxf(deesser_gr, st.deesser_gain_reduction_db);
    tp_events += st.true_peak_limited_events;
    for (size_t i = 0; i < len; ++i) {
      output_square_sum += (double)block[i] * (double)block[i];
      output_samples += 1;
    }
    if (out_audio) memcpy(out_audio + start, block, sizeof(float) * len);
  }

  float input_rms = input_samples > 0 ? (float)sqrt(input_square_sum / (double)input_samples) : 0.0f;
  float output_rms = output_samples > 0 ? (float)sqrt(output_square_sum / (double)output_samples) : 0.0f;
  float output_sample_peak_db = afo_linear_to_db_f32(output_sample_peak);
  float pre_limiter_true_peak_db = afo_linear_to_db_f32(pre_limiter_true_peak);
  float output_true_peak_db = afo_linear_to_db_f32(output_true_peak);

  float *tmp = (float *)malloc(sizeof(float) * (row_count ? row_count : 1));
  float *tmp2 = (float *)malloc(sizeof(float) * (row_count ? row_count : 1));
  for (size_t i = 0; i < row_count; ++i) tmp[i] = rows[i].in_db;
  float input_floor_db = afo_percentile_f32(tmp, row_count, 0.20f);
  float input_p90_db = afo_percentile_f32(tmp, row_count, 0.90f);
  float active_threshold_db = fmaxf(fmaxf(input_floor_db + 6.0f, input_p90_db - 24.0f), -60.0f);

  size_t active_n = 0;
  for (size_t i = 0; i < row_count; ++i)
    if (rows[i].in_db >= active_threshold_db) {
      tmp[active_n] = fmaxf(rows[i].comp_gr, 0.0f);
      tmp2[active_n] = fmaxf(rows[i].deesser_gr, 0.0f);
      active_n += 1;
    }
  if (active_n < 3) {
    active_n = row_count;
    for (size_t i = 0; i < row_count; ++i) {
      tmp[i] = fmaxf(rows[i].comp_gr, 0.0f);
      tmp2[i] = fmaxf(rows[i].deesser_gr, 0.0f);
    }
  }
  size_t active_block_count = active_n;
  float compressor_active_ratio = 0.0f;
  if (active_block_count > 0) {
    size_t cnt = 0;
    for (size_t i = 0; i < active_n; ++i) cnt += tmp[i] >= 0.10f;
    compressor_active_ratio = (float)cnt / (float)active_block_count;
  }
  float comp_median = afo_percentile_f32(tmp, active_n, 0.50f);
  float comp_p95 = afo_percentile_f32(tmp, active_n, 0.95f);
  float deesser_median = afo_percentile_f32(tmp2, active_n, 0.50f);
  float deesser_p95 = afo_percentile_f32(tmp2, active_n, 0.95f);

  size_t k = 0;
  for (size_t i = 0; i < row_count; ++i)
    if (rows[i].in_db >= active_threshold_db && rows[i].in_db > -100.0f) tmp[k++] = rows[i].out_db - rows[i].in_db;
  float active_output_gain_db = afo_percentile_f32(tmp, k, 0.50f);
  k = 0;
  for (size_t i = 0; i < row_count; ++i)
    if (rows[i].in_db < active_threshold_db && rows[i].in_db > -100.0f) tmp[k++] = rows[i].out_db - rows[i].in_db;
  float silence_level_delta_db = afo_percentile_f32(tmp, k, 0.50f);
  k = 0;
  for (size_t i = 0; i < row_count; ++i)
    if (rows[i].in_db < active_threshold_db) tmp[k++] = -fmaxf(rows[i].comp_gr, 0.0f);
  float silence_output_gain_db = afo_percentile_f32(tmp, k, 0.50f);
  for (size_t i = 0; i < row_count; ++i) tmp[i] = fmaxf(rows[i].comp_gr, 0.0f);
  float pumping = afo_pumping_score(tmp, row_count, 50.0f);

  memset(r, 0, sizeof(*r));
  r->input_sample_peak_db = afo_linear_to_db_f32(input_sample_peak);
  r->input_rms_db = afo_linear_to_db_f32(input_rms);
  r->output_sample_peak_db = output_sample_peak_db;
  r->pre_limiter_true_peak_db = pre_limiter_true_peak_db;
  r->output_true_peak_db = output_true_peak_db;
  r->output_rms_db = afo_linear_to_db_f32(output_rms);
  r->limiter_effective_ceiling_db = effective_ceiling_db;
  r->sample_headroom_db = effective_ceiling_db - output_sample_peak_db;
  r->pre_limiter_true_peak_headroom_db = effective_ceiling_db - pre_limiter_true_peak_db;
  r->true_peak_headroom_db = effective_ceiling_db - output_true_peak_db;
  r->limiter_gain_reduction_db = limiter_gr;
  r->true_peak_limiter_gain_reduction_db = tp_gr;
  r->true_peak_limited_events = tp_events;
  r->compressor_gain_reduction_db = comp_gr;
  r->deesser_gain_reduction_db = deesser_gr;
  r->compressor_gain_reduction_median_db = comp_median;
  r->compressor_gain_reduction_p95_db = comp_p95;
  r->compressor_gain_reduction_active_ratio = compressor_active_ratio;
  r->active_output_gain_db = active_output_gain_db;
  r->silence_output_gain_db = silence_output_gain_db;
  r->silence_level_delta_db = silence_level_delta_db;
  r->compressor_pumping_score_db = pumping;
  r->non_finite_output = non_finite_output;
  r->deesser_gain_reduction_median_db = deesser_median;
  r->deesser_gain_reduction_p95_db = deesser_p95;
  r->analysis_block_ms = 20.0f;
  r->active_analysis_threshold_db = active_threshold_db;
  r->active_analysis_block_count = active_block_count;
  r->processed_samples = output_samples;

  free(tmp); free(tmp2); free(rows); free(block);
  afo_chain_free(p);
  return 0;
}

/* ---------------------------------------------------------- simulate_eq_v2 */
/* lib.rs:214-288 */
int afo_simulate_eq_v2(const float *audio, size_t n, double sample_rate,
                       const afo_eq_band_config bands[AFO_NUM_BANDS], afo_eq_v2_result *r,
                       float *out_audio) {
  char msg[160];
  if (!isfinite(sample_rate) || sample_rate <= 0.0) return -1;
  for (size_t i = 0; i < AFO_NUM_BANDS; ++i)
    if (afo_eq_band_config_validate(&bands[i], i, sample_rate, msg, sizeof msg)) return -2;
  for (size_t i = 0; i < n; ++i)
    if (!isfinite(audio[i])) return -3;
  afo_eq eq;
  afo_eq_init(&eq, sample_rate);
  for (size_t i = 0; i < AFO_NUM_BANDS; ++i) afo_eq_set_band_config(&eq, i, &bands[i]);
  afo_eq_reset(&eq);
  float *output = (float *)malloc(sizeof(float) * (n ? n : 1));
  memcpy(output, audio, sizeof(float) * n);
  afo_eq_process_block(&eq, output, n);
  double in_sq = 0.0, out_sq = 0.0;
  float in_peak = 0.0f, out_peak = 0.0f;
  int non_finite = 0;
  for (size_t i = 0; i < n; ++i) {
    in_sq += (double)audio[i] * (double)audio[i];
    out_sq += (double)output[i] * (double)output[i];
    in_peak = fmaxf(in_peak, fabsf(audio[i]));
    out_peak = fmaxf(out_peak, fabsf(output[i]));
    non_finite |= !isfinite(output[i]);
  }
  double divisor = (double)(n > 1 ? n : 1);
  afo_tp_detector din, dout;
  afo_tp_detector_init(&din);
  afo_tp_detector_init(&dout);
  r->input_true_peak = afo_tp_detector_process_block(&din, audio, n);
  r->output_true_peak = afo_tp_detector_process_block(&dout, output, n);
  double max_response = -INFINITY;
  for (int i = 0; i < 512; ++i) {
    double f = 20.0 * pow(20000.0 / 20.0, (double)i / 511.0);
    double v;
    afo_eq_magnitude_response_db(&eq, &f, 1, &v);
    max_response = fmax(max_response, v);
  }
  r->input_sample_peak = in_peak;
  r->output_sample_peak = out_peak;
  r->input_rms = sqrt(in_sq / divisor);
  r->output_rms = sqrt(out_sq / divisor);
  r->max_response_db = max_response;
  r->sample_count = n;
  r->non_finite_output = non_finite;
  if (out_audio) memcpy(out_audio, output, sizeof(float) * n);
  free(output);
  return 0;
}

/* lib.rs:99-150 */
int afo_eq_magnitude_response(const double *freqs, size_t n, const double bands[AFO_NUM_BANDS][3],
                              double sample_rate, double *out) {
  if (!isfinite(sample_rate) || sample_rate <= 0.0) return -1;
  double nyquist = sample_rate / 2.0;
  for (size_t i = 0; i < AFO_NUM_BANDS; ++i) {
    double f = bands[i][0], g = bands[i][1], q = bands[i][2];
    if (!isfinite(f) || f <= 0.0 || f >= nyquist) return -2;
    if (!isfinite(g)) return -2;
    if (!isfinite(q) || q <= 0.0) return -2;
  }
  for (size_t i = 0; i < n; ++i)
    if (!isfinite(freqs[i]) || freqs[i] < 0.0 || freqs[i] > nyquist) return -3;
  afo_eq eq;
  afo_eq_init(&eq, sample_rate);
  for (size_t i = 0; i < AFO_NUM_BANDS; ++i) {
    afo_eq_set_band_frequency(&eq, i, bands[i][0]);
    afo_eq_set_band_gain(&eq, i, bands[i][1]);
    afo_eq_set_band_q(&eq, i, bands[i][2]);
  }
  afo_eq_magnitude_response_db(&eq, freqs, n, out);
  return 0;
}

/* lib.rs:191-212 */
int afo_eq_magnitude_response_v2(const double *freqs, size_t n,
                                 const afo_eq_band_config bands[AFO_NUM_BANDS], double sample_rate,
                                 double *out) {
  char msg[160];
  if (!isfinite(sample_rate) || sample_rate <= 0.0) return -1;
  for (size_t i = 0; i < AFO_NUM_BANDS; ++i)
    if (afo_eq_band_config_validate(&bands[i], i, sample_rate, msg, sizeof msg)) return -2;
  double nyquist = sample_rate / 2.0;
  for (size_t i = 0; i < n; ++i)
    if (!isfinite(freqs[i]) || freqs[i] < 0.0 || freqs[i] > nyquist) return -3;
  afo_eq eq;
  afo_eq_init(&eq, sample_rate);
  for (size_t i = 0; i < AFO_NUM_BANDS; ++i) afo_eq_set_band_config(&eq, i, &bands[i]);
  afo_eq_magnitude_response_db(&eq, freqs, n, out);
  return 0;
}

/* ------------------------------------------------------- KAT test signal */
/* audio/processor/tests.rs:1824-1851; fundamental 180 Hz / phrase 1.7 Hz is the KAT */
void afo_kat_signal(float *out, size_t n_blocks, uint64_t noise_state, double f0, double phrase_hz) {
  const double sample_rate = 48000.0;
  for (size_t block_index = 0; block_index < n_blocks; ++block_index) {
    for (size_t local = 0; local < 480; ++local) {
      size_t index = block_index * 480 + local;
      double time = (double)index / sample_rate;
      double phrase = 0.25 + 0.75 * fabs(sin(2.0 * M_PI * phrase_hz * time));
      double sibilant_gate = ((block_index / 12) % 5 == 2) ? 1.0 : 0.0;
      noise_state = noise_state * 6364136223846793005ULL + 1442695040888963407ULL;
      double noise =
          ((double)(uint32_t)(noise_state >> 40) / (double)((1u << 24) - 1) * 2.0 - 1.0) * 0.012;
      out[index] = (float)(phrase * (0.30 * sin(2.0 * M_PI * f0 * time) +
                                     0.14 * sin(2.0 * M_PI * (2.0 * f0) * time) +
                                     0.08 * sin(2.0 * M_PI * (15.0 * f0) * time)) +
                           sibilant_gate * 0.35 * sin(2.0 * M_PI * 7200.0 * time) + noise);
    }
  }
}

/* ------------------------------------------- simulate_auto_makeup_control */
/* audio/processor/python_api.rs:118-276.  traces: [6][block_count] f32 rows =
 * makeup_gain_db, activity, reliability, gain_reduction_db, input_rms_db, output_rms_db */
int afo_simulate_auto_makeup_control(const float *audio, size_t n, double sample_rate,
                                     const double *vad_probabilities, size_t n_vad,
                                     double noise_floor_db, double noise_reliability,
                                     const afo_makeup_settings *s, float *traces, float *out_audio) {
  const size_t CONTROL_BLOCK_SIZE = 480;
  if (!isfinite(sample_rate) || sample_rate <= 0.0) return -1;
  if (!isfinite(noise_floor_db) || !isfinite(noise_reliability) || noise_reliability < 0.0 ||
      noise_reliability > 1.0)
    return -2;
  for (size_t i = 0; i < n_vad; ++i)
    if (!isfinite(vad_probabilities[i]) || vad_probabilities[i] < 0.0 || vad_probabilities[i] > 1.0) return -3;
  size_t block_count = (n + CONTROL_BLOCK_SIZE - 1) / CONTROL_BLOCK_SIZE;
  if (n_vad != 0 && n_vad != block_count) return -4;
  if (!isfinite(s->vad_reliability) || s->vad_reliability < 0.0 || s->vad_reliability > 1.0) return -5;
  afo_compressor c;
  afo_compressor_init(&c, s->threshold_db, s->ratio, s->attack_ms, s->release_ms, s->makeup_gain_db, 6.0,
                      sample_rate);
  afo_compressor_set_auto_makeup_enabled(&c, 1);
  afo_compressor_set_target_lufs(&c, s->target_lufs);
  afo_compressor_set_noise_reference_reliability(&c, noise_reliability);
  afo_compressor_set_adaptive_release(&c, s->adaptive_release);
  afo_compressor_set_sidechain_highpass_enabled(&c, s->sidechain_highpass_enabled);
  float block[480];
  for (size_t b = 0; b < block_count; ++b) {
    size_t start = b * CONTROL_BLOCK_SIZE;
    size_t len = n - start < CONTROL_BLOCK_SIZE ? n - start : CONTROL_BLOCK_SIZE;
    memcpy(block, audio + start, sizeof(float) * len);
    double sq = 0.0;
    for (size_t i = 0; i < len; ++i) sq += (double)block[i] * (double)block[i];
    float input_rms = (float)sqrt(sq / (double)(len > 1 ? len : 1));
    afo_auto_makeup_input ev;
    const afo_auto_makeup_input *evp = NULL;
    if (b < n_vad) {
      ev.vad_probability = vad_probabilities[b];
      ev.vad_reliability = s->vad_reliability;
      ev.noise_floor_db = noise_floor_db;
      ev.live_noise_reliability = noise_reliability;
      evp = &ev;
    }
    afo_compressor_process_block(&c, block, len, evp);
    sq = 0.0;
    for (size_t i = 0; i < len; ++i) sq += (double)block[i] * (double)block[i];
    float output_rms = (float)sqrt(sq / (double)(len > 1 ? len : 1));
    traces[0 * block_count + b] = (float)c.smoothed_makeup_gain;
    traces[1 * block_count + b] = (float)c.speech_activity_score;
    traces[2 * block_count + b] = (float)c.auto_makeup_activity_reliability;
    traces[3 * block_count + b] = (float)c.current_gain_reduction_db;
    traces[4 * block_count + b] = afo_linear_to_db_f32(input_rms);
    traces[5 * block_count + b] = afo_linear_to_db_f32(output_rms);
    if (out_audio) memcpy(out_audio + start, block, sizeof(float) * len);
  }
  afo_compressor_free(&c);
  return 0;
}

/* lib.rs:290-298 over dsp/loudness.rs:43-83: ebur128 `Mode::I | Mode::HISTOGRAM`, mono (crate not vendored:
 * restated from the published libebur128 design the crate ports; parity unpinned).
 *   - K-weighted signal, 400 ms gating blocks every 100 ms ((rate + 5) / 10 frames);
 *   - HISTOGRAM mode: a block above the absolute gate (-70 LUFS) is counted in one of 1000 bins of 0.1 LU and from
 *     then on represented by its bin's centre energy; relative gate = mean of those - 10 LU; the result is the
 *     mean bin energy from the first bin at or above the relative gate.
 * A block's energy is accumulated as four 100 ms partial sums added oldest first (the GPU kernel produces the
 * partial sums; libebur128 walks its ring in storage order -- the two differ in the last bit at most). */
static double hist_energy(int i) { return pow(10.0, ((double)i / 10.0 - 69.95 + 0.691) / 10.0); }
static double hist_boundary(int i) { return pow(10.0, ((double)i / 10.0 - 70.0 + 0.691) / 10.0); }
static size_t find_histogram_index(double energy) {
  size_t lo = 0, hi = 1000;
  do {
    size_t mid = (lo + hi) / 2;
    if (energy >= hist_boundary((int)mid)) lo = mid; else hi = mid;
  } while (hi - lo != 1);
  return lo;
}
/* gating over 100 ms partial sums of y^2 (n100 of them, each over s100 frames) */
int afo_gated_loudness_from_partial_sums(const double *part, size_t n100, size_t s100, double *lufs) {
  unsigned long counts[1000];
  memset(counts, 0, sizeof counts);
  const double frames = (double)(s100 * 4);
  for (size_t b = 0; b + 4 <= n100; ++b) {
    const double energy = (((part[b] + part[b + 1]) + part[b + 2]) + part[b + 3]) / frames;
    if (energy >= hist_boundary(0)) counts[find_histogram_index(energy)]++;
  }
  double rel = 0.0; unsigned long above = 0;
  for (int i = 0; i < 1000; ++i) { rel += (double)counts[i] * hist_energy(i); above += counts[i]; }
  if (!above) return -4;
  rel /= (double)above;
  rel *= pow(10.0, -10.0 / 10.0);
  size_t start;
  if (rel < hist_boundary(0)) start = 0;
  else { start = find_histogram_index(rel); if (rel > hist_energy((int)start)) ++start; }
  double gated = 0.0; above = 0;
  for (size_t i = start; i < 1000; ++i) { gated += (double)counts[i] * hist_energy((int)i); above += counts[i]; }
  if (!above) return -4;
  gated /= (double)above;
  *lufs = 10.0 * (log(gated) / log(10.0)) - 0.691;
  return isfinite(*lufs) ? 0 : -4;
}
int afo_measure_integrated_loudness(const float *audio, size_t n, uint32_t sample_rate, double *lufs) {
  afo_loudness m;
  if (afo_loudness_init(&m, sample_rate)) return -1;
  if (n == 0) { afo_loudness_free(&m); return -2; }
  for (size_t i = 0; i < n; ++i)
    if (!isfinite(audio[i])) { afo_loudness_free(&m); return -3; }
  const size_t s100 = (sample_rate + 5) / 10, n100 = n / s100;
  double *part = (double *)calloc(n100 ? n100 : 1, sizeof(double));
  double v[5] = {0, 0, 0, 0, 0};
  for (size_t i = 0; i < n100 * s100; ++i) {
    v[0] = (double)audio[i] - m.a[1] * v[1] - m.a[2] * v[2] - m.a[3] * v[3] - m.a[4] * v[4];
    const double y = m.b[0] * v[0] + m.b[1] * v[1] + m.b[2] * v[2] + m.b[3] * v[3] + m.b[4] * v[4];
    v[4] = v[3]; v[3] = v[2]; v[2] = v[1]; v[1] = v[0];
    part[i / s100] += y * y;
  }
  int rc = afo_gated_loudness_from_partial_sums(part, n100, s100, lufs);
  free(part);
  afo_loudness_free(&m);
  return rc;
}


/* ------------------------------------------------------------------ noise gate (dsp/gate.rs) */
#define GATE_EXPANDER_RATIO 4.0
#define GATE_EXPANDER_RANGE_DB 36.0
#define GATE_AUTO_RELAX_RANGE_DB 24.0
#define GATE_HYSTERESIS_DB 4.0
/* gate.rs:158-225 */
void afo_gate_init(afo_gate *g, double threshold_db, double attack_ms, double release_ms, double sample_rate) {
  memset(g, 0, sizeof(*g));
  g->threshold_db = threshold_db;
  g->attack_coeff = afo_time_constant_to_coeff(attack_ms, sample_rate);
  g->release_coeff = afo_time_constant_to_coeff(release_ms, sample_rate);
  g->rms_coeff = afo_time_constant_to_coeff(8.0, sample_rate);
  g->detector_level_db = -120.0;
  g->sample_rate = sample_rate;
  g->enabled = 1;
}
/* gate.rs:812-821 */
void afo_gate_set_vad_mode(afo_gate *g, int vad_mode) {
  g->vad_mode = vad_mode;
  if (!vad_mode) g->auto_relax_remaining_samples = 0;
}
/* gate.rs:562-575 */
static void gate_advance_chatter_timers(afo_gate *g) {
  if (g->auto_relax_remaining_samples > 0) g->auto_relax_remaining_samples -= 1;
  if (g->chatter_window_remaining_samples > 0) {
    g->chatter_window_remaining_samples -= 1;
    if (g->chatter_window_remaining_samples == 0) g->chatter_transition_count = 0;
  }
  if (g->chatter_cooldown_samples > 0) g->chatter_cooldown_samples -= 1;
}
/* gate.rs:578-611 */
static void gate_track_transition(afo_gate *g, int effective_open) {
  if (!g->has_effective_gate_state) {
    g->effective_gate_open = effective_open;
    g->has_effective_gate_state = 1;
    gate_advance_chatter_timers(g);
    return;
  }
  if (effective_open != g->effective_gate_open) {
    g->effective_gate_open = effective_open;
    if (g->chatter_window_remaining_samples == 0) {
      g->chatter_window_remaining_samples = (size_t)round(g->sample_rate * 500.0 / 1000.0);
      g->chatter_transition_count = 1;
    } else {
      g->chatter_transition_count += 1;
    }
    if (g->chatter_transition_count >= 4 && g->chatter_cooldown_samples == 0) {
      g->chatter_event_count += 1;
      g->chatter_cooldown_samples = (size_t)round(g->sample_rate * 1000.0 / 1000.0);
      if (g->vad_mode) g->auto_relax_remaining_samples = (size_t)round(g->sample_rate * 700.0 / 1000.0);
      g->chatter_window_remaining_samples = 0;
      g->chatter_transition_count = 0;
    }
  }
  gate_advance_chatter_timers(g);
}
/* gate.rs:626-637 over update_detector (265-285), detector_gain_reduction_db (298-306), apply_gain (613-623) */
float afo_gate_process_sample(afo_gate *g, float input) {
  if (!g->enabled) return input;
  const double x = (double)input;
  g->rms_envelope_sq = g->rms_coeff * g->rms_envelope_sq + (1.0 - g->rms_coeff) * x * x;
  g->detector_level_db = afo_linear_to_db(sqrt(g->rms_envelope_sq), 1e-10);
  if (g->detector_level_db >= g->threshold_db) {
    g->is_open = 1;
    g->hold_remaining_samples = (size_t)round(g->sample_rate * 50.0 / 1000.0);
  } else if (g->hold_remaining_samples > 0) {
    g->hold_remaining_samples -= 1;
    g->is_open = 1;
  } else if (g->detector_level_db <= g->threshold_db - GATE_HYSTERESIS_DB) {
    g->is_open = 0;
  }
  const double range = g->auto_relax_remaining_samples > 0 ? GATE_AUTO_RELAX_RANGE_DB : GATE_EXPANDER_RANGE_DB;
  const double gr = g->is_open ? 0.0
                               : clampd((g->threshold_db - g->detector_level_db) * (1.0 - 1.0 / GATE_EXPANDER_RATIO), 0.0, range);
  gate_track_transition(g, g->is_open);
  const double target_gain = afo_db_to_linear(-gr);
  const double coeff = target_gain > g->current_gain ? g->attack_coeff : g->release_coeff;
  g->current_gain = coeff * g->current_gain + (1.0 - coeff) * target_gain;
  return (float)(x * g->current_gain);
}
void afo_gate_process_block(afo_gate *g, float *buf, size_t n) {
  if (!g->enabled) return;
  for (size_t i = 0; i < n; ++i) buf[i] = afo_gate_process_sample(g, buf[i]);
}
