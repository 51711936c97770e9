// af_host.hpp -- host-side mirror of the reference's stage structs: configuration only.
//
// The reference configures one `OfflineDspBlockProcessor` through per-stage setters whose
// side effects (coefficient crossfades, envelope resets, coupled release times) decide
// the arithmetic that follows.  This mirror reproduces those setters on a single
// prototype (no audio ever flows through it); `export_*` then flattens the prototype into
// the uniform ChainParams block and the initial per-stream state the kernels start from.
// Time constants and RBJ coefficients are computed here with the host libm -- the same
// functions the reference's `f64::exp/powf/sin/cos` lower to.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "af_device.h"

namespace af {

constexpr double kPi = 3.14159265358979323846264338327950288;

inline double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// dsp/util.rs:5-20
inline double time_constant_to_coeff(double time_ms, double sample_rate) {
  const double tau = std::fmax(time_ms, 0.001) / 1000.0;
  return std::exp(-1.0 / (tau * sample_rate));
}
inline double db_to_linear(double db) { return std::pow(10.0, db / 20.0); }

// ------------------------------------------------------------------------------ biquad
enum class BiquadType { LowShelf, HighShelf, Peaking, Notch, HighPass, LowPass, Bypass };

// dsp/biquad.rs:110-182
inline BiquadCoef rbj_coefficients(BiquadType type, double frequency, double gain_db, double q_in,
                                   double sample_rate) {
  const double omega = 2.0 * kPi * frequency / sample_rate;
  const double sn = std::sin(omega), cs = std::cos(omega);
  const double q = std::fmax(q_in, 1e-6);
  const double alpha = sn / (2.0 * q);
  const double a = std::pow(10.0, gain_db / 40.0);
  double b0, b1, b2, a0, a1, a2;
  switch (type) {
    case BiquadType::Peaking:
      b0 = 1.0 + alpha * a; b1 = -2.0 * cs; b2 = 1.0 - alpha * a;
      a0 = 1.0 + alpha / a; a1 = -2.0 * cs; a2 = 1.0 - alpha / a;
      break;
    case BiquadType::LowShelf: {
      const double t = 2.0 * std::sqrt(a) * alpha;
      b0 = a * ((a + 1.0) - (a - 1.0) * cs + t);
      b1 = 2.0 * a * ((a - 1.0) - (a + 1.0) * cs);
      b2 = a * ((a + 1.0) - (a - 1.0) * cs - t);
      a0 = (a + 1.0) + (a - 1.0) * cs + t;
      a1 = -2.0 * ((a - 1.0) + (a + 1.0) * cs);
      a2 = (a + 1.0) + (a - 1.0) * cs - t;
      break;
    }
    case BiquadType::HighShelf: {
      const double t = 2.0 * std::sqrt(a) * alpha;
      b0 = a * ((a + 1.0) + (a - 1.0) * cs + t);
      b1 = -2.0 * a * ((a - 1.0) + (a + 1.0) * cs);
      b2 = a * ((a + 1.0) + (a - 1.0) * cs - t);
      a0 = (a + 1.0) - (a - 1.0) * cs + t;
      a1 = 2.0 * ((a - 1.0) - (a + 1.0) * cs);
      a2 = (a + 1.0) - (a - 1.0) * cs - t;
      break;
    }
    case BiquadType::Notch:
      b0 = 1.0; b1 = -2.0 * cs; b2 = 1.0;
      a0 = 1.0 + alpha; a1 = -2.0 * cs; a2 = 1.0 - alpha;
      break;
    case BiquadType::HighPass:
      b0 = (1.0 + cs) / 2.0; b1 = -(1.0 + cs); b2 = (1.0 + cs) / 2.0;
      a0 = 1.0 + alpha; a1 = -2.0 * cs; a2 = 1.0 - alpha;
      break;
    case BiquadType::LowPass:
      b0 = (1.0 - cs) / 2.0; b1 = 1.0 - cs; b2 = (1.0 - cs) / 2.0;
      a0 = 1.0 + alpha; a1 = -2.0 * cs; a2 = 1.0 - alpha;
      break;
    default:
      b0 = 1.0; b1 = 0.0; b2 = 0.0; a0 = 1.0; a1 = 0.0; a2 = 0.0;
      break;
  }
  return BiquadCoef{b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0};
}

// dsp/biquad.rs:184-205
inline double coef_magnitude_db(const BiquadCoef &c, double frequency_hz, double sample_rate) {
  const double omega = 2.0 * kPi * frequency_hz / sample_rate;
  const double c1 = std::cos(omega), s1 = std::sin(omega);
  const double c2 = std::cos(2.0 * omega), s2 = std::sin(2.0 * omega);
  const double nr = c.b0 + c.b1 * c1 + c.b2 * c2;
  const double ni = -c.b1 * s1 - c.b2 * s2;
  const double dr = 1.0 + c.a1 * c1 + c.a2 * c2;
  const double di = -c.a1 * s1 - c.a2 * s2;
  const double np = nr * nr + ni * ni;
  const double dp = dr * dr + di * di;
  const double magnitude = std::sqrt(np / std::fmax(dp, 1.0e-30));
  return 20.0 * std::log10(std::fmax(magnitude, 1.0e-10));
}

// The configuration half of dsp/biquad.rs:39-66.  z1/z2 of the prototype stay 0 because
// nothing is processed on the host, so a scheduled crossfade starts both paths from 0.
struct BiquadProto {
  BiquadType type = BiquadType::Bypass;
  double frequency = 0, gain_db = 0, q = 1, sample_rate = 48000;
  BiquadCoef active{1, 0, 0, 0, 0}, pending{1, 0, 0, 0, 0};
  int xf_total = 0, xf_remaining = 0;

  BiquadProto() = default;
  BiquadProto(BiquadType t, double f, double g, double qq, double fs)
      : type(t), frequency(f), gain_db(g), q(qq), sample_rate(fs) {
    set_immediate(target());
  }
  BiquadCoef target() const { return rbj_coefficients(type, frequency, gain_db, q, sample_rate); }
  // biquad.rs:232-247
  void set_immediate(const BiquadCoef &c) {
    active = c;
    pending = c;
    xf_total = 0;
    xf_remaining = 0;
  }
  // biquad.rs:12-19, 249-260
  void schedule(const BiquadCoef &c) {
    pending = c;
    const double samples = std::round(sample_rate * 1.5 / 1000.0);
    int n = 1;
    if (std::isfinite(samples)) n = (int)std::min(4096.0, std::max(1.0, samples));
    xf_total = n;
    xf_remaining = n;
  }
  void reset() { set_immediate(target()); }                                  // biquad.rs:341-347
  void set_frequency(double f) { frequency = f; schedule(target()); }        // biquad.rs:350-353
  void set_q(double v) { q = std::fmax(v, 1e-6); schedule(target()); }       // biquad.rs:371-374
  void set_gain_db_immediate(double g) { gain_db = g; set_immediate(target()); }  // :365-368
  void set_parameters(BiquadType t, double f, double g, double qq) {         // biquad.rs:377-389
    type = t; frequency = f; gain_db = g; q = std::fmax(qq, 1e-6);
    schedule(target());
  }
  void set_parameters_immediate(BiquadType t, double f, double g, double qq) {  // :395-407
    type = t; frequency = f; gain_db = g; q = std::fmax(qq, 1e-6);
    set_immediate(target());
  }
  double target_magnitude_db(double hz) const { return coef_magnitude_db(target(), hz, sample_rate); }
  SectionParams section() const { return SectionParams{active, pending, xf_total, xf_remaining}; }
};

// ---------------------------------------------------------------------------------- EQ
struct EqBandConfig {
  int filter_type;  // AF_EQ_*
  double frequency_hz, gain_db, q;
  int slope_db_per_octave;
  bool enabled;
};

constexpr double kDefaultFrequencies[kNumBands] = {80.0,   160.0,  320.0,  640.0,   1280.0,
                                                   2500.0, 5000.0, 8000.0, 12000.0, 16000.0};
constexpr double kDefaultQ = 1.41;

inline bool eq_is_pass(int t) { return t == 4 || t == 5; }
inline bool slope_supported(int s) { return s == 12 || s == 24 || s == 36 || s == 48; }

// eq.rs:140-201; returns "" when valid
inline std::string eq_validate(const EqBandConfig &c, int index, double sample_rate) {
  char buf[200];
  const double fmin = 20.0;
  auto fail = [&](const char *fmt, auto... args) {
    if constexpr (sizeof...(args) == 0) {
      std::snprintf(buf, sizeof buf, "%s", fmt);
    } else {
      std::snprintf(buf, sizeof buf, fmt, args...);
    }
    return std::string("Band ") + std::to_string(index) + ": " + buf;
  };
  if (!std::isfinite(c.frequency_hz)) return fail("frequency must be finite");
  if (!std::isfinite(sample_rate) || sample_rate <= 2.0 * fmin)
    return fail("sample rate must be finite and support the EQ frequency range");
  const double fmax_hz = std::fmax(sample_rate / 2.0 - 1.0, fmin);
  if (!(c.frequency_hz >= fmin && c.frequency_hz <= fmax_hz))
    return fail("frequency %g Hz out of range [%g, %g]", c.frequency_hz, fmin, fmax_hz);
  if (!std::isfinite(c.gain_db)) return fail("gain must be finite");
  if (!(c.gain_db >= -12.0 && c.gain_db <= 12.0)) return fail("gain %g dB out of range [-12, 12]", c.gain_db);
  if (!std::isfinite(c.q)) return fail("Q must be finite");
  if (!(c.q >= 0.1 && c.q <= 10.0)) return fail("Q %g out of range [0.1, 10]", c.q);
  if (!slope_supported(c.slope_db_per_octave))
    return fail("slope %d dB/octave is unsupported; expected one of [12, 24, 36, 48]", c.slope_db_per_octave);
  return "";
}

struct EqBandProto {
  BiquadProto sections[kMaxSectionsPerBand];
  EqBandConfig config{};
  int processing_sections = 0, target_sections = 0;

  static BiquadType biquad_type(int t) {  // eq.rs:97-107
    switch (t) {
      case 0: return BiquadType::LowShelf;
      case 1: return BiquadType::Peaking;
      case 2: return BiquadType::HighShelf;
      case 3: return BiquadType::Notch;
      case 4: return BiquadType::HighPass;
      default: return BiquadType::LowPass;
    }
  }
  static int required_sections(const EqBandConfig &c) {  // eq.rs:248-256
    if (!c.enabled) return 0;
    if (eq_is_pass(c.filter_type)) return slope_supported(c.slope_db_per_octave) ? c.slope_db_per_octave / 12 : 1;
    return 1;
  }
  static void section_parameters(const EqBandConfig &c, int index, int count, BiquadType &t, double &g,
                                 double &q) {  // eq.rs:258-277, 203-207
    t = biquad_type(c.filter_type);
    if (eq_is_pass(c.filter_type)) {
      const int order = 2 * count;
      const double angle = (double)(2 * index + 1) * kPi / (double)(2 * order);
      g = 0.0;
      q = 1.0 / (2.0 * std::cos(angle));
    } else {
      g = c.filter_type == 3 ? 0.0 : c.gain_db;
      q = c.q;
    }
  }
  void init(const EqBandConfig &c, double fs) {  // eq.rs:223-246
    const int target = required_sections(c);
    for (int s = 0; s < kMaxSectionsPerBand; ++s) {
      if (s < target) {
        BiquadType t; double g, q;
        section_parameters(c, s, target, t, g, q);
        sections[s] = BiquadProto(t, c.frequency_hz, g, q, fs);
      } else {
        sections[s] = BiquadProto(BiquadType::Bypass, c.frequency_hz, 0.0, kDefaultQ, fs);
      }
    }
    config = c;
    processing_sections = target_sections = target;
  }
  void set_config(const EqBandConfig &c) {  // eq.rs:279-298
    config = c;
    const int target = required_sections(c);
    const int processing = std::max(processing_sections, target);
    for (int s = 0; s < processing; ++s) {
      BiquadType t = BiquadType::Bypass; double g = 0.0, q = kDefaultQ;
      if (s < target) section_parameters(c, s, target, t, g, q);
      sections[s].set_parameters(t, c.frequency_hz, g, q);
    }
    processing_sections = processing;
    target_sections = target;
  }
  void reset() {  // eq.rs:324-336
    const int target = required_sections(config);
    for (int s = 0; s < kMaxSectionsPerBand; ++s) {
      BiquadType t = BiquadType::Bypass; double g = 0.0, q = kDefaultQ;
      if (s < target) section_parameters(config, s, target, t, g, q);
      sections[s].set_parameters_immediate(t, config.frequency_hz, g, q);
    }
    processing_sections = target_sections = target;
  }
};

struct EqProto {
  EqBandProto bands[kNumBands];
  bool enabled = true;
  double sample_rate = 48000;

  explicit EqProto(double fs = 48000.0) : sample_rate(fs) {  // eq.rs:357-368, 122-138
    for (int i = 0; i < kNumBands; ++i) {
      EqBandConfig c{i == 0 ? 0 : (i == 9 ? 2 : 1), kDefaultFrequencies[i], 0.0, kDefaultQ, 12, true};
      bands[i].init(c, fs);
    }
  }
  void reset() { for (auto &b : bands) b.reset(); }
  void set_band_gain(int i, double g) { auto c = bands[i].config; c.gain_db = g; bands[i].set_config(c); }
  void set_band_frequency(int i, double f) { auto c = bands[i].config; c.frequency_hz = f; bands[i].set_config(c); }
  void set_band_q(int i, double q) { auto c = bands[i].config; c.q = q; bands[i].set_config(c); }
  void set_band_config(int i, const EqBandConfig &c) { bands[i].set_config(c); }
  // eq.rs:338-343, 511-527
  double magnitude_db(double hz) const {
    if (!enabled) return 0.0;
    double total = 0.0;
    for (const auto &b : bands) {
      double band_sum = 0.0;
      for (int s = 0; s < b.target_sections; ++s) band_sum += b.sections[s].target_magnitude_db(hz);
      total += band_sum;
    }
    return total;
  }
};

// ---------------------------------------------------------------------- K-weighting
// ebur128 0.1.10 is not vendored in the reference (Cargo.lock:236-245); this is the filter of
// ITU-R BS.1770-4 as the published libebur128 design builds it for an arbitrary rate: a high
// shelf (f0 1681.97 Hz, +4 dB) cascaded with the RLB high-pass (f0 38.14 Hz), expanded into one
// 4th-order transfer function.  Parity for this block is unpinned.
inline void kweighting_design(double fs, double b[5], double a[5]) {
  double f0 = 1681.974450955533, G = 3.999843853973347, Q = 0.7071752369554196;
  double K = std::tan(kPi * f0 / fs);
  const double Vh = std::pow(10.0, G / 20.0), Vb = std::pow(Vh, 0.4996667741545416);
  double pb[3], pa[3] = {1.0, 0.0, 0.0};
  const double rb[3] = {1.0, -2.0, 1.0};
  double ra[3] = {1.0, 0.0, 0.0};
  const double a0 = 1.0 + K / Q + K * K;
  pb[0] = (Vh + Vb * K / Q + K * K) / a0;
  pb[1] = 2.0 * (K * K - Vh) / a0;
  pb[2] = (Vh - Vb * K / Q + K * K) / a0;
  pa[1] = 2.0 * (K * K - 1.0) / a0;
  pa[2] = (1.0 - K / Q + K * K) / a0;
  f0 = 38.13547087602444;
  Q = 0.5003270373238773;
  K = std::tan(kPi * f0 / fs);
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

// --------------------------------------------------------------------------- compressor
// Parameter half + the state the setters touch, dsp/compressor.rs:131-404.
struct CompressorProto {
  double threshold_db, ratio, attack_coeff, release_coeff, detector_release_coeff;
  double makeup_gain_db, knee_db, rms_coeff, sample_rate;
  bool enabled = true, adaptive_release = false;
  double base_release_ms, current_release_ms, target_release_ms, release_smoothing_coeff;
  double current_gain_reduction_db = 0.0, fast_release_env_db = 0.0, slow_release_env_db = 0.0;
  bool auto_makeup_enabled = false, has_meter = true;
  double target_lufs = -18.0, smoothed_makeup_gain, makeup_smoothing_coeff;
  double speech_activity_smoothing_coeff, makeup_silence_relax_coeff, noise_reference_reliability = 0.0;
  bool sidechain_highpass_enabled = false;
  double sidechain_highpass_coeff;

  CompressorProto(double thr, double rat, double attack_ms, double release_ms, double makeup, double knee,
                  double fs) {  // compressor.rs:133-202
    sample_rate = fs;
    threshold_db = thr;
    ratio = std::fmax(rat, 1.0);
    attack_coeff = time_constant_to_coeff(attack_ms, fs);
    release_coeff = time_constant_to_coeff(release_ms, fs);
    detector_release_coeff = release_coeff;
    makeup_gain_db = makeup;
    knee_db = std::fmax(knee, 0.0);
    rms_coeff = time_constant_to_coeff(20.0, fs);
    base_release_ms = current_release_ms = target_release_ms = release_ms;
    release_smoothing_coeff = time_constant_to_coeff(100.0, fs);
    smoothed_makeup_gain = makeup;
    makeup_smoothing_coeff = time_constant_to_coeff(200.0, fs);
    speech_activity_smoothing_coeff = time_constant_to_coeff(200.0, fs);
    makeup_silence_relax_coeff = time_constant_to_coeff(1500.0, fs);
    const double cutoff = clampd(120.0, 20.0, fs * 0.45);  // compressor.rs:390-394
    sidechain_highpass_coeff = 1.0 / (1.0 + 2.0 * kPi * cutoff / std::fmax(fs, 1.0));
    const unsigned ok[] = {8000, 16000, 32000, 44100, 48000, 88200, 96000};  // loudness.rs:36-41
    has_meter = std::find(std::begin(ok), std::end(ok), (unsigned)fs) != std::end(ok);
  }
  void set_threshold(double v) { threshold_db = v; fast_release_env_db = current_gain_reduction_db; slow_release_env_db = 0.0; }
  void set_ratio(double v) { ratio = std::fmax(v, 1.0); }
  void set_attack_time(double ms) { attack_coeff = time_constant_to_coeff(ms, sample_rate); }
  void set_release_time(double ms) {  // compressor.rs:236-244
    base_release_ms = ms;
    if (!adaptive_release) {
      current_release_ms = target_release_ms = ms;
      release_coeff = time_constant_to_coeff(ms, sample_rate);
    }
    detector_release_coeff = time_constant_to_coeff(ms, sample_rate);
  }
  void set_adaptive_release(bool on) {  // compressor.rs:247-260
    adaptive_release = on;
    if (!on) {
      current_release_ms = target_release_ms = base_release_ms;
      release_coeff = time_constant_to_coeff(current_release_ms, sample_rate);
    }
    fast_release_env_db = current_gain_reduction_db;
    slow_release_env_db = 0.0;
  }
  void set_base_release_time(double ms) {  // compressor.rs:268-275
    base_release_ms = ms;
    if (!adaptive_release) {
      current_release_ms = target_release_ms = ms;
      release_coeff = time_constant_to_coeff(ms, sample_rate);
    }
  }
  void set_makeup_gain(double db) { makeup_gain_db = db; if (!auto_makeup_enabled) smoothed_makeup_gain = db; }
  void set_auto_makeup_enabled(bool on) { auto_makeup_enabled = on && has_meter; if (!on) smoothed_makeup_gain = makeup_gain_db; }
  void set_target_lufs(double v) { target_lufs = clampd(v, -24.0, -12.0); }
  void set_sidechain_highpass_enabled(bool on) { sidechain_highpass_enabled = on; }  // state is still all-zero
  void set_noise_reference_reliability(double v) { noise_reference_reliability = std::isfinite(v) ? clampd(v, 0.0, 1.0) : 0.0; }

  CompressorParams params(int control_block) const {
    CompressorParams p{};
    kweighting_design(sample_rate, p.kw_b, p.kw_a);
    const int s100 = ((int)sample_rate + 5) / 10;
    p.meter_frames = 4.0 * s100;
    p.meter_slots = (has_meter && control_block > 0 && (4 * s100) % control_block == 0 && (4 * s100) / control_block <= 64)
                        ? (4 * s100) / control_block : 0;
    p.makeup_pow_cb = std::pow(makeup_smoothing_coeff, (double)control_block);
    p.relax_pow_cb = std::pow(makeup_silence_relax_coeff, (double)control_block);
    p.activity_pow_cb = std::pow(speech_activity_smoothing_coeff, (double)control_block);
    p.threshold_db = threshold_db; p.ratio = ratio; p.knee_db = knee_db;
    p.attack_coeff = attack_coeff; p.detector_release_coeff = detector_release_coeff; p.rms_coeff = rms_coeff;
    p.release_smoothing_coeff = release_smoothing_coeff; p.base_release_ms = base_release_ms;
    p.band_env_coeff = time_constant_to_coeff(18.0, sample_rate);
    p.fast_release_coeff = time_constant_to_coeff(50.0, sample_rate);
    p.slow_charge_coeff = time_constant_to_coeff(250.0, sample_rate);
    p.slow_release_coeff = time_constant_to_coeff(400.0, sample_rate);
    p.sidechain_highpass_coeff = sidechain_highpass_coeff;
    p.makeup_gain_db = makeup_gain_db; p.makeup_smoothing_coeff = makeup_smoothing_coeff;
    p.makeup_silence_relax_coeff = makeup_silence_relax_coeff;
    p.speech_activity_smoothing_coeff = speech_activity_smoothing_coeff;
    p.target_lufs = target_lufs; p.noise_reference_reliability = noise_reference_reliability;
    p.sample_rate = sample_rate;
    p.comp_factor = 1.0 - 1.0 / ratio;
    p.knee_start = threshold_db - knee_db / 2.0;
    p.knee_end = threshold_db + knee_db / 2.0;
    p.two_knee = 2.0 * knee_db;
    p.two_knee_recip = knee_db > 0.0 ? 1.0 / (2.0 * knee_db) : 0.0;
    p.adaptive_release = adaptive_release; p.sidechain_highpass_enabled = sidechain_highpass_enabled;
    p.auto_makeup_enabled = auto_makeup_enabled;
    return p;
  }
};

// ------------------------------------------------------------------------------ limiter
struct LimiterProto {  // dsp/limiter.rs:106-184
  double ceiling_db, ceiling_linear, release_coeff, sample_rate;
  int lookahead_samples;
  bool enabled = true;
  static int samples_for(double ms, double fs) {
    const double s = std::round(clampd(ms, 0.1, 10.0) / 1000.0 * fs);
    return (int)std::min((double)kMaxLookahead, std::max(1.0, s));
  }
  LimiterProto(double ceil_db, double release_ms, double fs, double lookahead_ms)
      : ceiling_db(ceil_db), ceiling_linear(db_to_linear(ceil_db)),
        release_coeff(time_constant_to_coeff(release_ms, fs)), sample_rate(fs),
        lookahead_samples(samples_for(lookahead_ms, fs)) {}
  void set_ceiling(double db) { ceiling_db = std::fmin(db, 0.0); ceiling_linear = db_to_linear(ceiling_db); }
  void set_release_time(double ms) { release_coeff = time_constant_to_coeff(ms, sample_rate); }
  void set_lookahead_ms(double ms) { lookahead_samples = samples_for(ms, sample_rate); }
  LimiterParams params() const { return LimiterParams{ceiling_db, ceiling_linear, release_coeff, lookahead_samples, 0}; }
};

// --------------------------------------------------------------------- true-peak limiter
struct TruePeakProto {  // dsp/true_peak.rs:266-313
  float ceiling_linear, release_coeff, sample_rate;
  TruePeakProto(float fs, float ceiling_db, float release_ms) {
    ceiling_linear = (float)db_to_linear((double)ceiling_db);
    sample_rate = std::fmax(fs, 1.0f);
    set_release_ms(release_ms);
  }
  void set_ceiling_linear(float c) { ceiling_linear = clampf(c, 0.000001f, 1.0f); }
  void set_release_ms(float ms) {
    release_coeff = (float)time_constant_to_coeff((double)clampf(ms, 5.0f, 500.0f), (double)sample_rate);
  }
};

// ----------------------------------------------------------------------------- de-esser
struct DeEsserBandProto {
  double low_hz, high_hz;
  BiquadProto detector_hp, detector_lp, dynamic_eq;
};
struct DeEsserProto {  // dsp/deesser.rs:109-353
  bool enabled = false, auto_enabled = true;
  double auto_amount = 0.5, threshold_db = -28.0, ratio = 4.0, max_reduction_db = 6.0;
  double attack_coeff, release_coeff, detector_attack_coeff, detector_release_coeff;
  double low_cut_hz = 4000.0, high_cut_hz = 11000.0, sample_rate;
  DeEsserBandProto bands[3];

  static double center_hz(double lo, double hi) { return std::sqrt(lo * hi); }
  static double dyn_q(double lo, double hi) { return clampd(center_hz(lo, hi) / std::fmax(hi - lo, 200.0), 0.5, 6.0); }
  explicit DeEsserProto(double fs) : sample_rate(fs) {
    attack_coeff = time_constant_to_coeff(2.0, fs);
    release_coeff = time_constant_to_coeff(80.0, fs);
    detector_attack_coeff = time_constant_to_coeff(1.5, fs);
    detector_release_coeff = time_constant_to_coeff(60.0, fs);
    double lo[3], hi[3];
    bounds(lo, hi);
    for (int i = 0; i < 3; ++i) {
      bands[i].low_hz = lo[i];
      bands[i].high_hz = hi[i];
      bands[i].detector_hp = BiquadProto(BiquadType::HighPass, lo[i], 0.0, 0.707, fs);
      bands[i].detector_lp = BiquadProto(BiquadType::LowPass, hi[i], 0.0, 0.707, fs);
      bands[i].dynamic_eq = BiquadProto(BiquadType::Peaking, center_hz(lo[i], hi[i]), 0.0, dyn_q(lo[i], hi[i]), fs);
    }
  }
  void bounds(double lo[3], double hi[3]) const {  // deesser.rs:231-261
    const double span = std::fmax(high_cut_hz - low_cut_hz, 600.0);
    const double a = low_cut_hz + span / 3.0, b = low_cut_hz + span * 2.0 / 3.0;
    lo[0] = low_cut_hz; hi[0] = a; lo[1] = a; hi[1] = b; lo[2] = b; hi[2] = high_cut_hz;
  }
  void rebuild() {  // deesser.rs:65-74, 231-245
    double lo[3], hi[3];
    bounds(lo, hi);
    for (int i = 0; i < 3; ++i) {
      bands[i].low_hz = lo[i];
      bands[i].high_hz = hi[i];
      bands[i].detector_hp.set_frequency(lo[i]);
      bands[i].detector_lp.set_frequency(hi[i]);
      bands[i].dynamic_eq.set_frequency(center_hz(lo[i], hi[i]));
      bands[i].dynamic_eq.set_q(dyn_q(lo[i], hi[i]));
    }
  }
  void set_auto_amount(double v) { auto_amount = clampd(v, 0.0, 1.0); }
  void set_low_cut_hz(double v) {
    low_cut_hz = clampd(v, 2000.0, 12000.0);
    if (high_cut_hz <= low_cut_hz + 200.0) high_cut_hz = clampd(low_cut_hz + 200.0, 2200.0, 16000.0);
    rebuild();
  }
  void set_high_cut_hz(double v) {
    high_cut_hz = clampd(v, 2200.0, 16000.0);
    if (high_cut_hz <= low_cut_hz + 200.0) low_cut_hz = clampd(high_cut_hz - 200.0, 2000.0, 12000.0);
    rebuild();
  }
  void set_threshold_db(double v) { threshold_db = clampd(v, -60.0, -6.0); }
  void set_ratio(double v) { ratio = clampd(v, 1.0, 20.0); }
  void set_attack_ms(double v) { attack_coeff = time_constant_to_coeff(clampd(v, 0.1, 50.0), sample_rate); }
  void set_release_ms(double v) { release_coeff = time_constant_to_coeff(clampd(v, 5.0, 500.0), sample_rate); }
  void set_max_reduction_db(double v) { max_reduction_db = clampd(v, 0.0, 24.0); }

  DeEsserParams params() const {
    DeEsserParams p{};
    p.attack_coeff = attack_coeff;
    p.release_coeff = release_coeff;
    p.detector_attack_coeff = detector_attack_coeff;
    p.detector_release_coeff = detector_release_coeff;
    p.max_reduction_db = max_reduction_db;
    p.threshold_db = threshold_db;
    p.ratio = ratio;
    p.auto_amount = auto_amount;
    p.baseline_fall = time_constant_to_coeff(13.88, sample_rate);
    p.baseline_rise = time_constant_to_coeff(34.72, sample_rate);
    p.baseline_inactive = time_constant_to_coeff(20.82, sample_rate);
    p.auto_enabled = auto_enabled;
    for (int i = 0; i < 3; ++i) {
      p.bands[i].detector_hp = bands[i].detector_hp.section();
      p.bands[i].detector_lp = bands[i].detector_lp.section();
      p.bands[i].dynamic_eq = bands[i].dynamic_eq.section();
      const BiquadProto &d = bands[i].dynamic_eq;
      const double omega = 2.0 * kPi * d.frequency / d.sample_rate;
      p.bands[i].dyn_cos_omega = std::cos(omega);
      p.bands[i].dyn_alpha = std::sin(omega) / (2.0 * std::fmax(d.q, 1e-6));
    }
    return p;
  }
};

// ------------------------------------------------------------------- the whole prototype
struct ChainProto {  // audio/processor/block_processor.rs:31-60
  double sample_rate;
  DeEsserProto deesser;
  EqProto eq;
  CompressorProto compressor;
  LimiterProto limiter;
  TruePeakProto tp_limiter;
  bool deesser_enabled = false, eq_enabled = true, compressor_enabled = false, limiter_enabled = true;
  bool eq_before_deesser = false;
  bool input_scrub = true, input_clamp = false, dc_block = false, pre_highpass = false;
  int control_block = 960;

  explicit ChainProto(double fs)
      : sample_rate(fs), deesser(fs), eq(fs), compressor(-18.0, 3.0, 5.0, 100.0, 0.0, 6.0, fs),
        limiter(-0.5, 50.0, fs, 2.0), tp_limiter((float)fs, -1.5f, 80.0f) {
    control_block = (int)std::min(8192.0, std::max(1.0, std::round(fs * 0.020)));  // python_api.rs:512-514
  }
};

}  // namespace af
