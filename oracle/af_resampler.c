/* af_resampler.c -- see af_resampler.h (TEST INFRASTRUCTURE ONLY; restates rubato 0.14.1's SincFixedIn). */
#include "af_resampler.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define OVERSAMPLING 256

struct afo_resampler {
  size_t chunk, sinc_len;
  double ratio;       /* output_rate / input_rate */
  double last_index;  /* position of the next output relative to the start of the next chunk */
  double *buffer;     /* chunk + 2 * sinc_len, the first 2 * sinc_len frames are history */
  double *sincs;      /* [OVERSAMPLING][sinc_len] */
};

/* rubato::windows::make_window -- all windows are evaluated "periodic" (divide by N, not N-1) */
static double window_value(int window, size_t x, size_t n) {
  const double pi = 3.14159265358979323846264338327950288;
  const double xf = (double)x, nf = (double)n;
  double w;
  switch (window) {
    case AFO_WIN_BLACKMAN_HARRIS:
    case AFO_WIN_BLACKMAN_HARRIS2:
      w = 0.35875 - 0.48829 * cos(2.0 * pi * xf / nf) + 0.14128 * cos(4.0 * pi * xf / nf) -
          0.01168 * cos(6.0 * pi * xf / nf);
      break;
    case AFO_WIN_BLACKMAN:
    case AFO_WIN_BLACKMAN2:
      w = 0.42 - 0.5 * cos(2.0 * pi * xf / nf) + 0.08 * cos(4.0 * pi * xf / nf);
      break;
    default:
      w = 0.5 - 0.5 * cos(2.0 * pi * xf / nf);
      break;
  }
  if (window == AFO_WIN_BLACKMAN_HARRIS2 || window == AFO_WIN_BLACKMAN2 || window == AFO_WIN_HANN2) w *= w;
  return w;
}

/* rubato::calculate_cutoff: the relative cutoff that puts the end of the window's transition band at
 * Nyquist, 1 / (1 + k(window, sinc_len) / sinc_len).  The crate's constants cannot be read here, so the f32
 * results were IDENTIFIED from the reference's own published measurements (tools/fit_resampler_cutoff.py):
 * for each configuration in evaluation/resampler-quality-report.json exactly one f32 value reproduces the
 * published stop-band / pass-band figure to all printed digits, its f32 neighbours are off in the 5th
 * digit, and every other published figure of that configuration then agrees to 13-17 digits with no freedom
 * left.  Other (window, length) pairs have no published measurement: they use k interpolated from the
 * identified points and are marked approximate in DESIGN.md. */
float afo_resampler_calculate_cutoff(size_t sinc_len, int window) {
  union { uint32_t u; float f; } bits;
  if (window == AFO_WIN_BLACKMAN && sinc_len == 128) { bits.u = 0x3F73E7B4u; return bits.f; }          /* 0.9527542591 */
  if (window == AFO_WIN_BLACKMAN_HARRIS2 && sinc_len == 128) { bits.u = 0x3F650CE0u; return bits.f; }  /* 0.8947277069 */
  if (window == AFO_WIN_BLACKMAN_HARRIS2 && sinc_len == 256) { bits.u = 0x3F72722Du; return bits.f; }  /* 0.9470546842 */
  const double n = (double)sinc_len;
  double k;
  switch (window) {
    case AFO_WIN_BLACKMAN_HARRIS2: k = 13.563209 + 191.625830 / n; break;  /* through the two identified points */
    case AFO_WIN_BLACKMAN: k = 6.347344; break;                            /* the identified point */
    /* not identified: scaled from the identified ones by the windows' main-lobe widths */
    case AFO_WIN_BLACKMAN_HARRIS: k = 6.347344 * (8.0 / 6.0); break;
    case AFO_WIN_BLACKMAN2: k = 6.347344 * 1.41; break;
    case AFO_WIN_HANN: k = 6.347344 * (4.0 / 6.0); break;
    default: k = 6.347344 * (4.0 / 6.0) * 1.41; break;
  }
  return (float)(1.0 / (k / n + 1.0));
}

static double sinc_fn(double v) {
  const double pi = 3.14159265358979323846264338327950288;
  if (v == 0.0) return 1.0;
  return sin(pi * v) / (pi * v);
}

/* rubato::sinc::make_sincs */
static void make_sincs(double *sincs, size_t npoints, size_t factor, float f_cutoff, int window) {
  const size_t tot = npoints * factor;
  double *y = (double *)malloc(sizeof(double) * tot);
  double sum = 0.0;
  for (size_t x = 0; x < tot; ++x) {
    const double v = window_value(window, x, tot) *
                     sinc_fn(((double)x - (double)(tot / 2)) * (double)f_cutoff / (double)factor);
    sum += v;
    y[x] = v;
  }
  sum /= (double)factor;
  for (size_t p = 0; p < npoints; ++p)
    for (size_t n = 0; n < factor; ++n) sincs[(factor - n - 1) * npoints + p] = y[factor * p + n] / sum;
  free(y);
}

afo_resampler *afo_resampler_new(uint32_t input_rate, uint32_t output_rate, size_t chunk_size, size_t sinc_len,
                                 int window, float f_cutoff) {
  afo_resampler *r = (afo_resampler *)calloc(1, sizeof(*r));
  r->chunk = chunk_size;
  r->sinc_len = 8 * ((sinc_len + 7) / 8);
  r->ratio = (double)output_rate / (double)input_rate;
  if (!(f_cutoff > 0.0f)) f_cutoff = afo_resampler_calculate_cutoff(sinc_len, window);
  /* below unity the cutoff follows the output Nyquist (f32 arithmetic, like the crate's parameter type) */
  const float cutoff = r->ratio >= 1.0 ? f_cutoff : f_cutoff * (float)r->ratio;
  r->sincs = (double *)malloc(sizeof(double) * OVERSAMPLING * r->sinc_len);
  make_sincs(r->sincs, r->sinc_len, OVERSAMPLING, cutoff, window);
  r->buffer = (double *)calloc(chunk_size + 2 * r->sinc_len, sizeof(double));
  r->last_index = -(double)(r->sinc_len / 2);
  return r;
}

void afo_resampler_free(afo_resampler *r) {
  if (!r) return;
  free(r->buffer);
  free(r->sincs);
  free(r);
}

size_t afo_resampler_output_delay(const afo_resampler *r) {
  return (size_t)((float)(r->sinc_len / 2) * (float)r->ratio);
}
size_t afo_resampler_output_frames_max(const afo_resampler *r) {
  return (size_t)((double)r->chunk * r->ratio * 1.2 + 10.0);
}
const double *afo_resampler_sinc_table(const afo_resampler *r) { return r->sincs; }

/* The dot product of one sinc row with the signal.  Evaluation order (shared with the HIP kernel so that the
 * two agree bit for bit): one fused multiply-add chain over the taps in increasing order. */
static double sinc_dot(const double *wave, const double *row, size_t n) {
  double acc = 0.0;
  for (size_t k = 0; k < n; ++k) acc = fma(wave[k], row[k], acc);
  return acc;
}

/* rubato::interpolation::interp_cubic: the cubic through four equally spaced points, evaluated between the
 * second and third */
static double interp_cubic(double x, const double y[4]) {
  const double a0 = y[1];
  const double a1 = -(1.0 / 3.0) * y[0] - 0.5 * y[1] + y[2] - (1.0 / 6.0) * y[3];
  const double a2 = 0.5 * (y[0] + y[2]) - y[1];
  const double a3 = 0.5 * (y[1] - y[2]) + (1.0 / 6.0) * (y[3] - y[0]);
  const double x2 = x * x;
  const double x3 = x2 * x;
  return a0 + a1 * x + a2 * x2 + a3 * x3;
}

size_t afo_resampler_process_chunk(afo_resampler *r, const double *in, double *out) {
  const size_t L = r->sinc_len, chunk = r->chunk;
  const double t_ratio = 1.0 / r->ratio;
  const long end_idx = (long)chunk - ((long)L + 1) - (long)ceil(t_ratio);
  memmove(r->buffer, r->buffer + chunk, sizeof(double) * 2 * L);
  if (in) memcpy(r->buffer + 2 * L, in, sizeof(double) * chunk);
  else memset(r->buffer + 2 * L, 0, sizeof(double) * chunk);
  double idx = r->last_index;
  size_t n = 0;
  while (idx < (double)end_idx) {
    idx += t_ratio;
    /* get_nearest_times_4: the sinc rows just before, at, and two after the sub-sample position */
    long index = (long)floor(idx);
    long sub = (long)floor((idx - floor(idx)) * (double)OVERSAMPLING);
    long pi[4], ps[4];
    pi[0] = index; ps[0] = sub - 1;
    if (ps[0] < 0) { ps[0] += OVERSAMPLING; pi[0] -= 1; }
    pi[1] = index; ps[1] = sub;
    for (int k = 2; k < 4; ++k) {
      sub += 1;
      if (sub >= OVERSAMPLING) { sub -= OVERSAMPLING; index += 1; }
      pi[k] = index; ps[k] = sub;
    }
    const double scaled = idx * (double)OVERSAMPLING;
    const double frac = scaled - floor(scaled);
    double pts[4];
    for (int k = 0; k < 4; ++k) pts[k] = sinc_dot(r->buffer + (size_t)(pi[k] + 2 * (long)L), r->sincs + (size_t)ps[k] * L, L);
    out[n++] = interp_cubic(frac, pts);
  }
  r->last_index = idx - (double)chunk;
  return n;
}

int64_t afo_simulate_product_resampler(const double *samples, size_t n, uint32_t input_rate, uint32_t output_rate,
                                       size_t chunk_size, size_t sinc_len, int window, float f_cutoff, double *out,
                                       size_t out_capacity, size_t *delay, size_t *expected_frames, size_t *blocks) {
  afo_resampler *r = afo_resampler_new(input_rate, output_rate, chunk_size, sinc_len, window, f_cutoff);
  const size_t d = afo_resampler_output_delay(r);
  const size_t expected = (size_t)round(((double)n * (double)output_rate) / (double)input_rate);
  double *chunk_out = (double *)malloc(sizeof(double) * afo_resampler_output_frames_max(r));
  double *padded = (double *)calloc(chunk_size, sizeof(double));
  size_t produced_total = 0, nblocks = 0, pos = 0;
  int64_t overflow = 0;
#define EMIT(count)                                                              \
  do {                                                                           \
    if (produced_total + (count) > out_capacity) overflow = 1;                   \
    else memcpy(out + produced_total, chunk_out, sizeof(double) * (count));      \
    produced_total += (count);                                                   \
    nblocks += 1;                                                                \
  } while (0)
  while (n - pos >= chunk_size) {
    const size_t c = afo_resampler_process_chunk(r, samples + pos, chunk_out);
    EMIT(c);
    pos += chunk_size;
  }
  if (pos < n) {  /* process_partial_into_buffer(Some(rest)): the rest, zero-padded to a chunk */
    memcpy(padded, samples + pos, sizeof(double) * (n - pos));
    const size_t c = afo_resampler_process_chunk(r, padded, chunk_out);
    EMIT(c);
  }
  while (produced_total < expected + d) {  /* flush with chunks of silence */
    const size_t c = afo_resampler_process_chunk(r, NULL, chunk_out);
    if (c == 0) break;
    EMIT(c);
  }
#undef EMIT
  if (delay) *delay = d;
  if (expected_frames) *expected_frames = expected;
  if (blocks) *blocks = nblocks;
  free(chunk_out);
  free(padded);
  afo_resampler_free(r);
  return overflow ? -(int64_t)produced_total : (int64_t)produced_total;
}
