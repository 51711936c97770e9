// af_resampler_host.hpp -- host side of the batched product resampler.
//
// The reference resamples 44.1 kHz-origin streams with rubato 0.14.1's asynchronous windowed-sinc resampler
// (rust-core/src/audio/processor/resampling.rs:140-156: sinc_len 128, Blackman window, 256 oversampled sinc
// rows, cubic interpolation between four neighbouring rows, chunks of 1024) and drives it with the loop of
// `simulate_product_resampler` (resampling.rs:179-261).  Where every output sample sits on the input time
// axis depends only on (ratio, chunk size, sinc_len) -- never on the audio -- so the host replays the
// reference's chunk loop ONCE per job into a table of per-output positions, and the GPU then evaluates all
// streams x all outputs as an embarrassingly parallel table-driven FIR (af_resampler.hip).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace af {

constexpr int kResampleOversampling = 256;
constexpr int kResampleTablePad = 16;  // zero taps either side of every sinc row (see af_resampler.hip)

enum ResampleWindow : int { kWinBlackmanHarris = 0, kWinBlackmanHarris2, kWinBlackman, kWinBlackman2, kWinHann, kWinHann2 };

// One output sample: the four (input offset, sinc row) pairs around its sub-sample position and the
// fractional position between the middle two (rubato get_nearest_times_4 + interp_cubic).
struct ResamplePos {
  int64_t base;     // absolute input index of the earliest of the four windows (may be negative: zeros)
  double frac;
  uint16_t sub[4];  // sinc row of each point
  uint8_t off[4];   // window start of each point relative to `base` (0..2)
  uint8_t pad[4];
};
static_assert(sizeof(ResamplePos) == 32, "position record");

// rubato::calculate_cutoff as identified from the reference's published measurements
// (tools/fit_resampler_cutoff.py; the three measured configurations are exact f32 values).
inline float resample_calculate_cutoff(int sinc_len, int window) {
  auto from_bits = [](uint32_t u) { float f; std::memcpy(&f, &u, sizeof f); return f; };
  if (window == kWinBlackman && sinc_len == 128) return from_bits(0x3F73E7B4u);
  if (window == kWinBlackmanHarris2 && sinc_len == 128) return from_bits(0x3F650CE0u);
  if (window == kWinBlackmanHarris2 && sinc_len == 256) return from_bits(0x3F72722Du);
  const double n = (double)sinc_len;
  double k;
  switch (window) {
    case kWinBlackmanHarris2: k = 13.563209 + 191.625830 / n; break;
    case kWinBlackman: k = 6.347344; break;
    case kWinBlackmanHarris: k = 6.347344 * (8.0 / 6.0); break;  // not identified: main-lobe scaling
    case kWinBlackman2: k = 6.347344 * 1.41; break;
    case kWinHann: k = 6.347344 * (4.0 / 6.0); break;
    default: k = 6.347344 * (4.0 / 6.0) * 1.41; break;
  }
  return (float)(1.0 / (k / n + 1.0));
}

struct ResamplePlan {
  uint32_t input_rate = 0, output_rate = 0;
  int64_t chunk = 1024;
  int sinc_len = 128, window = kWinBlackman;
  double ratio = 1.0;
  std::vector<double> table;  // [256][sinc_len + 2 * pad], zero padded rows

  int row_stride() const { return sinc_len + 2 * kResampleTablePad; }

  void build(uint32_t in_rate, uint32_t out_rate, int64_t chunk_size, int len, int win) {
    input_rate = in_rate;
    output_rate = out_rate;
    chunk = chunk_size;
    sinc_len = 8 * ((len + 7) / 8);
    window = win;
    ratio = (double)out_rate / (double)in_rate;
    const float fc = resample_calculate_cutoff(len, win);
    const float cutoff = ratio >= 1.0 ? fc : fc * (float)ratio;
    // windowed sinc over sinc_len * 256 points, normalised to unit DC gain per row set
    const size_t tot = (size_t)sinc_len * kResampleOversampling;
    const double pi = 3.14159265358979323846264338327950288;
    std::vector<double> y(tot);
    double sum = 0.0;
    for (size_t x = 0; x < tot; ++x) {
      const double xf = (double)x, nf = (double)tot;
      double w;
      switch (win) {
        case kWinBlackmanHarris:
        case kWinBlackmanHarris2:
          w = 0.35875 - 0.48829 * std::cos(2.0 * pi * xf / nf) + 0.14128 * std::cos(4.0 * pi * xf / nf) -
              0.01168 * std::cos(6.0 * pi * xf / nf);
          break;
        case kWinBlackman:
        case kWinBlackman2:
          w = 0.42 - 0.5 * std::cos(2.0 * pi * xf / nf) + 0.08 * std::cos(4.0 * pi * xf / nf);
          break;
        default:
          w = 0.5 - 0.5 * std::cos(2.0 * pi * xf / nf);
          break;
      }
      if (win == kWinBlackmanHarris2 || win == kWinBlackman2 || win == kWinHann2) w *= w;
      const double v = (xf - (double)(tot / 2)) * (double)cutoff / (double)kResampleOversampling;
      const double s = v == 0.0 ? 1.0 : std::sin(pi * v) / (pi * v);
      y[x] = w * s;
      sum += y[x];
    }
    sum /= (double)kResampleOversampling;
    table.assign((size_t)kResampleOversampling * row_stride(), 0.0);
    for (int p = 0; p < sinc_len; ++p)
      for (int n = 0; n < kResampleOversampling; ++n)
        table[(size_t)(kResampleOversampling - n - 1) * row_stride() + kResampleTablePad + p] =
            y[(size_t)kResampleOversampling * p + n] / sum;
  }

  int output_delay() const { return (int)((float)(sinc_len / 2) * (float)ratio); }
  int64_t expected_frames(int64_t n_in) const {
    return (int64_t)std::round(((double)n_in * (double)output_rate) / (double)input_rate);
  }

  // Replays the reference's driver loop (full chunks, one zero-padded partial chunk, silent flush chunks until
  // expected + delay frames exist) and records where every produced frame reads its input.
  int64_t positions(int64_t n_in, std::vector<ResamplePos> &pos, int64_t *blocks) const {
    pos.clear();
    const double t_ratio = 1.0 / ratio;
    const long end_idx = (long)chunk - ((long)sinc_len + 1) - (long)std::ceil(t_ratio);
    const int64_t target = expected_frames(n_in) + output_delay();
    double last_index = -(double)(sinc_len / 2);
    int64_t nblocks = 0, chunk_index = 0;
    auto run_chunk = [&]() {
      double idx = last_index;
      int64_t made = 0;
      while (idx < (double)end_idx) {
        idx += t_ratio;
        long index = (long)std::floor(idx);
        long sub = (long)std::floor((idx - std::floor(idx)) * (double)kResampleOversampling);
        long pi_[4], ps[4];
        pi_[0] = index; ps[0] = sub - 1;
        if (ps[0] < 0) { ps[0] += kResampleOversampling; pi_[0] -= 1; }
        pi_[1] = index; ps[1] = sub;
        for (int k = 2; k < 4; ++k) {
          sub += 1;
          if (sub >= kResampleOversampling) { sub -= kResampleOversampling; index += 1; }
          pi_[k] = index; ps[k] = sub;
        }
        const double scaled = idx * (double)kResampleOversampling;
        ResamplePos r{};
        r.base = chunk_index * chunk + pi_[0];
        r.frac = scaled - std::floor(scaled);
        for (int k = 0; k < 4; ++k) {
          r.sub[k] = (uint16_t)ps[k];
          r.off[k] = (uint8_t)(pi_[k] - pi_[0]);
        }
        pos.push_back(r);
        ++made;
      }
      last_index = idx - (double)chunk;
      ++chunk_index;
      ++nblocks;
      return made;
    };
    int64_t consumed = 0;
    while (n_in - consumed >= chunk) { run_chunk(); consumed += chunk; }
    if (consumed < n_in) run_chunk();
    while ((int64_t)pos.size() < target) {
      if (run_chunk() == 0) break;
    }
    if (blocks) *blocks = nblocks;
    return (int64_t)pos.size();
  }
};

}  // namespace af
