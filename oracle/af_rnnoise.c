/*
 * af_rnnoise.c -- CPU restatement of the RNNoise suppressor behind
 * rust-core/src/dsp/rnnoise.rs (TEST INFRASTRUCTURE ONLY, see af_oracle.h).
 *
 * PARITY UNPINNED.  The wrapper half (frame buffering, soft clip, x32768 scaling, wet/dry
 * smoothing: rnnoise.rs:45-164) is restated from the reference text.  The core,
 * `nnnoiseless::DenoiseState::process_frame` (Cargo.lock:605-613, call site rnnoise.rs:142-143),
 * is a third-party crate that is not vendored under /root/reference and whose trained weights
 * are embedded in the crate; what follows restates the published RNNoise algorithm that crate
 * ports (Valin, "A Hybrid DSP/Deep Learning Approach to Real-Time Full-Band Speech Enhancement":
 * 480-sample frames, 960-point Vorbis-window STFT, 22 Bark-like bands, 42 features with pitch
 * analysis, dense(42->24) -> GRU24 -> GRU48 -> GRU96 -> dense(96->22), pitch comb filter,
 * band-gain interpolation, overlap-add) and runs it on seeded synthetic int8 weights in the real
 * layer layout.  The reference's own tests for this stage assert only frame counts and finiteness
 * (rnnoise.rs:355-450), so nothing here can be pinned against golden data.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "af_rnnoise.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define FRAME AFO_RNN_FRAME
#define WINDOW 960
#define FREQ 481
#define PMIN 60
#define PMAX 768
#define PFRAME 960
#define PBUF 1728
#define NB AFO_RNN_BANDS
#define CEPS_MEM 8
#define NDELTA 6
#define NFEAT AFO_RNN_FEATURES

static const int eband5ms[NB] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40, 48, 60, 78, 100};

/* ------------------------------------------------------------- tables */
static float half_window[FRAME];
static float dct_table[NB * NB];
static float tansig_table[201];
static float tw_re[WINDOW], tw_im[WINDOW];
static int tables_ready = 0;

static void init_tables(void) {
  if (tables_ready) return;
  for (int i = 0; i < FRAME; ++i) {
    double s = sin(.5 * M_PI * (i + .5) / FRAME);
    half_window[i] = (float)sin(.5 * M_PI * s * s);
  }
  for (int i = 0; i < NB; ++i)
    for (int j = 0; j < NB; ++j) {
      double v = cos((i + .5) * j * M_PI / NB);
      if (j == 0) v *= sqrt(.5);
      dct_table[i * NB + j] = (float)v;
    }
  for (int i = 0; i <= 200; ++i) tansig_table[i] = (float)tanh(0.04 * i);
  for (int i = 0; i < WINDOW; ++i) {
    tw_re[i] = (float)cos(-2.0 * M_PI * i / WINDOW);
    tw_im[i] = (float)sin(-2.0 * M_PI * i / WINDOW);
  }
  tables_ready = 1;
}

/* --------------------------------------------------------------- FFT-960
 * Mixed-radix decimation-in-time (960 = 4*4*4*3*5), f32 arithmetic, twiddles rounded from f64.
 * out = (1/960) * sum_n in[n] exp(-2 pi i k n / 960)   (the forward scaling of the Opus FFT) */
typedef struct { float r, i; } cpx;

static void fft_rec(const cpx *in, cpx *out, int n, int stride) {
  if (n == 1) { out[0] = in[0]; return; }
  int p = (n % 4 == 0) ? 4 : (n % 3 == 0) ? 3 : 5;
  int m = n / p;
  cpx tmp[WINDOW];
  for (int q = 0; q < p; ++q) fft_rec(in + q * stride, tmp + q * m, m, stride * p);
  for (int k = 0; k < m; ++k) {
    for (int r = 0; r < p; ++r) {
      int kk = k + r * m; /* output bin */
      float sr = 0.0f, si = 0.0f;
      for (int q = 0; q < p; ++q) {
        int t = (int)(((long)q * kk * (WINDOW / n)) % WINDOW);
        float wr = tw_re[t], wi = tw_im[t];
        float xr = tmp[q * m + k].r, xi = tmp[q * m + k].i;
        sr += xr * wr - xi * wi;
        si += xr * wi + xi * wr;
      }
      out[kk].r = sr;
      out[kk].i = si;
    }
  }
}

/* ---- timed path (afo_rnn_fft_mode = 1): the same transforms as a 480-point complex mixed-radix FFT (4.4.2.3.5, decimation
 * in time, twiddles from the table above) over the even/odd packing of the real window.  The checker path above stays
 * what the parity tests were taken with; this one exists so that the CPU baseline of bench.py is not handicapped by a
 * plain recursive transform (results differ from the checker path in the last bits only: tests/test_oracle_rnnoise_wrapper.py). */
int afo_rnn_fft_mode = 0;
#define HALF 480
static const int fast_radix[5] = {4, 4, 2, 3, 5};

/* out[k], k < n: DFT of in[0], in[stride], ... (n = product of radix[0..]); recursion depth 5, no scratch */
static void fast_fft_rec(cpx *out, const cpx *in, int n, int stride, const int *radix) {
  const int p = radix[0], m = n / p;
  if (m == 1) {
    for (int q = 0; q < p; ++q) out[q] = in[q * stride];
  } else {
    for (int q = 0; q < p; ++q) fast_fft_rec(out + q * m, in + q * stride, m, stride * p, radix + 1);
  }
  const int tw_step = WINDOW / n; /* W_n^j = tw[j * 960 / n] */
  for (int k = 0; k < m; ++k) {
    cpx t[5] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
    for (int q = 0; q < p; ++q) {
      const cpx v = out[q * m + k];
      const int ti = q * k * tw_step;
      const float wr = tw_re[ti], wi = tw_im[ti];
      t[q].r = v.r * wr - v.i * wi;
      t[q].i = v.r * wi + v.i * wr;
    }
    if (p == 2) {
      out[k].r = t[0].r + t[1].r; out[k].i = t[0].i + t[1].i;
      out[k + m].r = t[0].r - t[1].r; out[k + m].i = t[0].i - t[1].i;
    } else if (p == 4) {
      const float ar = t[0].r + t[2].r, ai = t[0].i + t[2].i, br = t[0].r - t[2].r, bi = t[0].i - t[2].i;
      const float cr = t[1].r + t[3].r, ci = t[1].i + t[3].i, dr = t[1].r - t[3].r, di = t[1].i - t[3].i;
      out[k].r = ar + cr; out[k].i = ai + ci;
      out[k + m].r = br + di; out[k + m].i = bi - dr;           /* b - i d */
      out[k + 2 * m].r = ar - cr; out[k + 2 * m].i = ai - ci;
      out[k + 3 * m].r = br - di; out[k + 3 * m].i = bi + dr;   /* b + i d */
    } else if (p == 3) {
      const float c = 0.86602540378443864676f;
      const float sr = t[1].r + t[2].r, si = t[1].i + t[2].i, dr = t[1].r - t[2].r, di = t[1].i - t[2].i;
      const float mr = t[0].r - 0.5f * sr, mi = t[0].i - 0.5f * si;
      out[k].r = t[0].r + sr; out[k].i = t[0].i + si;
      out[k + m].r = mr + c * di; out[k + m].i = mi - c * dr;
      out[k + 2 * m].r = mr - c * di; out[k + 2 * m].i = mi + c * dr;
    } else { /* 5 */
      const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
      const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
      const float a1r = t[1].r + t[4].r, a1i = t[1].i + t[4].i, b1r = t[1].r - t[4].r, b1i = t[1].i - t[4].i;
      const float a2r = t[2].r + t[3].r, a2i = t[2].i + t[3].i, b2r = t[2].r - t[3].r, b2i = t[2].i - t[3].i;
      const float m1r = t[0].r + c1 * a1r + c2 * a2r, m1i = t[0].i + c1 * a1i + c2 * a2i;
      const float m2r = t[0].r + c2 * a1r + c1 * a2r, m2i = t[0].i + c2 * a1i + c1 * a2i;
      const float n1r = s1 * b1r + s2 * b2r, n1i = s1 * b1i + s2 * b2i;
      const float n2r = s2 * b1r - s1 * b2r, n2i = s2 * b1i - s1 * b2i;
      out[k].r = t[0].r + a1r + a2r; out[k].i = t[0].i + a1i + a2i;
      out[k + m].r = m1r + n1i; out[k + m].i = m1i - n1r;           /* m1 - i n1 */
      out[k + 4 * m].r = m1r - n1i; out[k + 4 * m].i = m1i + n1r;
      out[k + 2 * m].r = m2r + n2i; out[k + 2 * m].i = m2i - n2r;
      out[k + 3 * m].r = m2r - n2i; out[k + 3 * m].i = m2i + n2r;
    }
  }
}

static void fast_forward_transform(cpx *out /*FREQ*/, const float *in /*WINDOW*/) {
  cpx z[HALF], Z[HALF];
  for (int n = 0; n < HALF; ++n) { z[n].r = in[2 * n]; z[n].i = in[2 * n + 1]; }
  fast_fft_rec(Z, z, HALF, 1, fast_radix);
  const float s = 1.0f / WINDOW;
  for (int k = 0; k <= HALF; ++k) {
    const cpx a = Z[k % HALF], b = Z[(HALF - k) % HALF]; /* Z[k], Z[480 - k] */
    const float er = 0.5f * (a.r + b.r), ei = 0.5f * (a.i - b.i);   /* even part: (Z[k] + conj Z[N-k]) / 2 */
    const float orr = 0.5f * (a.i + b.i), oi = -0.5f * (a.r - b.r); /* odd part: (Z[k] - conj Z[N-k]) / (2 i) */
    const float wr = tw_re[k % WINDOW], wi = tw_im[k % WINDOW];     /* W_960^k */
    out[k].r = (er + (orr * wr - oi * wi)) * s;
    out[k].i = (ei + (orr * wi + oi * wr)) * s;
  }
}

static void fast_inverse_transform(float *out /*WINDOW*/, const cpx *in /*FREQ*/) {
  /* x[2n] + i x[2n+1] = sum_k Z[k] W_480^{-kn},  Z[k] = (X[k] + X[k+480]) + i (X[k] - X[k+480]) W_960^{-k},
   * X[k+480] = conj X[480-k]; the unscaled inverse runs through the forward kernel on conjugated input */
  cpx Zc[HALF], z[HALF];
  for (int k = 0; k < HALF; ++k) {
    const cpx a = in[k], b = in[HALF - k];
    const float sr = a.r + b.r, si = a.i - b.i;   /* X[k] + conj X[480-k] */
    const float dr = a.r - b.r, di = a.i + b.i;   /* X[k] - conj X[480-k] */
    const float wr = tw_re[k], wi = -tw_im[k];    /* W_960^{-k} */
    const float tr = dr * wr - di * wi, ti = dr * wi + di * wr;
    /* Z = s + i t; store conj(Z) */
    Zc[k].r = sr - ti;
    Zc[k].i = -(si + tr);
  }
  fast_fft_rec(z, Zc, HALF, 1, fast_radix);
  for (int n = 0; n < HALF; ++n) { out[2 * n] = z[n].r; out[2 * n + 1] = -z[n].i; }
}

static void forward_transform(cpx *out /*FREQ*/, const float *in /*WINDOW*/) {
  if (afo_rnn_fft_mode == 1) { fast_forward_transform(out, in); return; }
  cpx x[WINDOW], y[WINDOW];
  for (int i = 0; i < WINDOW; ++i) { x[i].r = in[i]; x[i].i = 0.0f; }
  fft_rec(x, y, WINDOW, 1);
  const float s = 1.0f / WINDOW;
  for (int i = 0; i < FREQ; ++i) { out[i].r = y[i].r * s; out[i].i = y[i].i * s; }
}

static void inverse_transform(float *out /*WINDOW*/, const cpx *in /*FREQ*/) {
  if (afo_rnn_fft_mode == 1) { fast_inverse_transform(out, in); return; }
  cpx x[WINDOW], y[WINDOW];
  for (int i = 0; i < FREQ; ++i) x[i] = in[i];
  for (int i = FREQ; i < WINDOW; ++i) { x[i].r = x[WINDOW - i].r; x[i].i = -x[WINDOW - i].i; }
  fft_rec(x, y, WINDOW, 1);
  /* forward FFT of a conjugate-symmetric spectrum; time reversal gives the inverse */
  out[0] = y[0].r;
  for (int i = 1; i < WINDOW; ++i) out[i] = y[WINDOW - i].r;
}

static void apply_window(float *x) {
  for (int i = 0; i < FRAME; ++i) {
    x[i] *= half_window[i];
    x[WINDOW - 1 - i] *= half_window[i];
  }
}

/* ------------------------------------------------------------ band tools */
/* Band b = R_b + F_b: R_b sums the rising ramp over band b-1's bins, F_b the falling ramp over band b's bins.
 * Evaluation order (the GPU kernels use the very same one): a band segment is cut into blocks of 8 bins, each
 * block is summed left to right, and the block sums are added in block order.  The scalar C of RNNoise uses
 * one running accumulator per band; this differs from it only in the last roundings. */
static void band_accumulate(float *bandE, const cpx *X, const cpx *P) {
  if (afo_rnn_eval_order == 1) { /* compute_band_energy / compute_band_corr of the published C: one accumulator per band */
    float sum[NB] = {0};
    for (int i = 0; i < NB - 1; ++i) {
      int band_size = (eband5ms[i + 1] - eband5ms[i]) << 2;
      for (int j = 0; j < band_size; ++j) {
        float frac = (float)j / band_size;
        int idx = (eband5ms[i] << 2) + j;
        float tmp = X[idx].r * P[idx].r + X[idx].i * P[idx].i;
        sum[i] += (1 - frac) * tmp;
        sum[i + 1] += frac * tmp;
      }
    }
    sum[0] *= 2;
    sum[NB - 1] *= 2;
    memcpy(bandE, sum, sizeof sum);
    return;
  }
  float rise[NB] = {0}, fall[NB] = {0};
  for (int i = 0; i < NB - 1; ++i) {
    int band_size = (eband5ms[i + 1] - eband5ms[i]) << 2;
    float rise_total = 0, fall_total = 0;
    for (int k0 = 0; k0 < band_size; k0 += 8) {
      float r = 0, f = 0;
      for (int j = k0; j < k0 + 8 && j < band_size; ++j) {
        float frac = (float)j / band_size;
        int idx = (eband5ms[i] << 2) + j;
        float tmp = X[idx].r * P[idx].r + X[idx].i * P[idx].i;
        f += (1 - frac) * tmp;
        r += frac * tmp;
      }
      rise_total += r;
      fall_total += f;
    }
    fall[i] = fall_total;
    rise[i + 1] = rise_total;
  }
  for (int i = 0; i < NB; ++i) bandE[i] = rise[i] + fall[i];
  bandE[0] *= 2;
  bandE[NB - 1] *= 2;
}
static void compute_band_energy(float *bandE, const cpx *X) { band_accumulate(bandE, X, X); }
static void compute_band_corr(float *bandE, const cpx *X, const cpx *P) { band_accumulate(bandE, X, P); }

static void interp_band_gain(float *g /*FREQ*/, const float *bandE) {
  memset(g, 0, sizeof(float) * FREQ);
  for (int i = 0; i < NB - 1; ++i) {
    int band_size = (eband5ms[i + 1] - eband5ms[i]) << 2;
    for (int j = 0; j < band_size; ++j) {
      float frac = (float)j / band_size;
      g[(eband5ms[i] << 2) + j] = (1 - frac) * bandE[i] + frac * bandE[i + 1];
    }
  }
}

static void dct(float *out, const float *in) {
  for (int i = 0; i < NB; ++i) {
    float sum = 0;
    for (int j = 0; j < NB; ++j) sum += in[j] * dct_table[j * NB + i];
    out[i] = sum * sqrtf(2.0f / 22);
  }
}

/* ------------------------------------------------------------ pitch tools
 * Evaluation order.  The scalar C of RNNoise sums long dot products left to right; the Rust crate the
 * reference uses unrolls them over several accumulators, so no single order is "the" reference order.
 * This restatement fixes one that a 64-lane wavefront evaluates natively, and the GPU kernels use the
 * very same one:
 *   dot64(x, y, n): 64 interleaved partial sums p[l] = sum_m x[l+64m]*y[l+64m] (mul then add, m ascending),
 *                   combined by the xor butterfly p[l] += p[l^32], ^16, ^8, ^4, ^2, ^1.
 *   scan64(e, n):   prefix sums with each lane owning ceil(n/64) consecutive terms: sequential prefix inside
 *                   a lane, Hillis-Steele inclusive scan of the 64 lane totals, one add of the offset.
 * The short lag-parallel correlations (coarse pitch search) keep the plain left-to-right order, each term one fused
 * multiply-add: that is what a matrix-core instruction evaluates, and the GPU runs them there. */
/* afo_rnn_eval_order: 0 = the wavefront-native order described above (what the GPU kernels evaluate);
 * 1 = the order of the published scalar C (one running accumulator, left to right, unfused multiply-add; band sums with one
 * accumulator per band; the yy table as a running sum).  Order 1 is implementation-independent: tests hold both the GPU and
 * order 0 against it and count the frames whose pitch decision differs (the crate itself is absent, so neither is "the"
 * reference order -- see the parity note at the top). */
int afo_rnn_eval_order = 0;

static float dot_seq(const float *x, const float *y, int n) {
  float s = 0;
  for (int i = 0; i < n; ++i) s = s + x[i] * y[i];
  return s;
}

static float dot64(const float *x, const float *y, int n) {
  if (afo_rnn_eval_order == 1) return dot_seq(x, y, n);
  float p[64];
  for (int l = 0; l < 64; ++l) p[l] = 0.0f;
  int i0 = 0;
  for (; i0 + 64 <= n; i0 += 64) /* p[l] takes its terms in ascending i either way; this loop order vectorises */
    for (int l = 0; l < 64; ++l) p[l] = p[l] + x[i0 + l] * y[i0 + l];
  for (int l = 0; i0 + l < n; ++l) p[l] = p[l] + x[i0 + l] * y[i0 + l];
  for (int off = 32; off >= 1; off >>= 1) {
    float q[64];
    for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
    memcpy(p, q, sizeof p);
  }
  return p[0];
}

/* out[i] = e[0] + ... + e[i] for i < n (n <= 64 * 8) in the blocked order described above */
static void scan64(const float *e, float *out, int n) {
  if (afo_rnn_eval_order == 1) {
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) { acc = acc + e[i]; out[i] = acc; }
    return;
  }
  const int chunk = (n + 63) / 64;
  float total[64], local[512];
  for (int l = 0; l < 64; ++l) {
    float acc = 0.0f;
    for (int k = 0; k < chunk; ++k) {
      const int i = l * chunk + k;
      if (i < n) {
        acc = acc + e[i];
        local[i] = acc;
      }
    }
    total[l] = acc;
  }
  for (int off = 1; off < 64; off <<= 1) {
    float q[64];
    for (int l = 0; l < 64; ++l) q[l] = l >= off ? total[l] + total[l - off] : total[l];
    memcpy(total, q, sizeof total);
  }
  for (int l = 0; l < 64; ++l)
    for (int k = 0; k < chunk; ++k) {
      const int i = l * chunk + k;
      if (i < n) out[i] = l == 0 ? local[i] : total[l - 1] + local[i];
    }
}

static void pitch_xcorr(const float *x, const float *y, float *xcorr, int len, int max_pitch) {
  if (afo_rnn_eval_order == 1) {
    for (int i = 0; i < max_pitch; ++i) xcorr[i] = dot_seq(x, y + i, len);
    return;
  }
  for (int i = 0; i < max_pitch; ++i) xcorr[i] = 0.0f;
  for (int j = 0; j < len; ++j) { /* every lag accumulates its terms in ascending j, one fused multiply-add each */
    const float xj = x[j];
    for (int i = 0; i < max_pitch; ++i) xcorr[i] = fmaf(xj, y[i + j], xcorr[i]);
  }
}

static void celt_lpc4(float *lpc, const float *ac) {
  const int p = 4;
  float error = ac[0];
  for (int i = 0; i < p; ++i) lpc[i] = 0;
  if (ac[0] != 0) {
    for (int i = 0; i < p; ++i) {
      float rr = 0;
      for (int j = 0; j < i; ++j) rr += lpc[j] * ac[i - j];
      rr += ac[i + 1];
      float r = -rr / error;
      lpc[i] = r;
      for (int j = 0; j < (i + 1) >> 1; ++j) {
        float t1 = lpc[j], t2 = lpc[i - 1 - j];
        lpc[j] = t1 + r * t2;
        lpc[i - 1 - j] = t2 + r * t1;
      }
      error = error - r * r * error;
      if (error < .001f * ac[0]) break;
    }
  }
}

static void pitch_downsample(const float *x, float *x_lp, int len) {
  int half = len >> 1;
  for (int i = 1; i < half; ++i) x_lp[i] = .5f * (.5f * (x[2 * i - 1] + x[2 * i + 1]) + x[2 * i]);
  x_lp[0] = .5f * (.5f * x[1] + x[0]);
  float ac[5];
  for (int k = 0; k <= 4; ++k) ac[k] = dot64(x_lp + k, x_lp, half - k);
  ac[0] *= 1.0001f;
  for (int i = 1; i <= 4; ++i) ac[i] -= ac[i] * (.008f * i) * (.008f * i);
  float lpc[4];
  celt_lpc4(lpc, ac);
  float tmp = 1.0f;
  for (int i = 0; i < 4; ++i) {
    tmp = .9f * tmp;
    lpc[i] = lpc[i] * tmp;
  }
  const float c1 = .8f;
  float n0 = lpc[0] + .8f, n1 = lpc[1] + c1 * lpc[0], n2 = lpc[2] + c1 * lpc[1], n3 = lpc[3] + c1 * lpc[2],
        n4 = c1 * lpc[3];
  float m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0;
  for (int i = 0; i < half; ++i) {
    float sum = x_lp[i];
    sum += n0 * m0;
    sum += n1 * m1;
    sum += n2 * m2;
    sum += n3 * m3;
    sum += n4 * m4;
    m4 = m3; m3 = m2; m2 = m1; m1 = m0; m0 = x_lp[i];
    x_lp[i] = sum;
  }
}

static void find_best_pitch(const float *xcorr, const float *y, int len, int max_pitch, int *best_pitch) {
  float Syy = 1;
  float best_num[2] = {-1, -1}, best_den[2] = {0, 0};
  best_pitch[0] = 0;
  best_pitch[1] = 1;
  Syy = 1.0f + dot64(y, y, len);
  for (int i = 0; i < max_pitch; ++i) {
    if (xcorr[i] > 0) {
      float xcorr16 = xcorr[i] * 1e-12f;
      float num = xcorr16 * xcorr16;
      if (num * best_den[1] > best_num[1] * Syy) {
        if (num * best_den[0] > best_num[0] * Syy) {
          best_num[1] = best_num[0]; best_den[1] = best_den[0]; best_pitch[1] = best_pitch[0];
          best_num[0] = num; best_den[0] = Syy; best_pitch[0] = i;
        } else {
          best_num[1] = num; best_den[1] = Syy; best_pitch[1] = i;
        }
      }
    }
    Syy += y[i + len] * y[i + len] - y[i] * y[i];
    Syy = fmaxf(1, Syy);
  }
}

static void pitch_search(const float *x_lp, const float *y, int len, int max_pitch, int *pitch) {
  int lag = len + max_pitch;
  float x_lp4[PFRAME >> 2], y_lp4[(PFRAME + PMAX) >> 2], xcorr[PMAX >> 1];
  int best_pitch[2] = {0, 0};
  for (int j = 0; j < len >> 2; ++j) x_lp4[j] = x_lp[2 * j];
  for (int j = 0; j < lag >> 2; ++j) y_lp4[j] = y[2 * j];
  pitch_xcorr(x_lp4, y_lp4, xcorr, len >> 2, max_pitch >> 2);
  find_best_pitch(xcorr, y_lp4, len >> 2, max_pitch >> 2, best_pitch);
  for (int i = 0; i < max_pitch >> 1; ++i) {
    xcorr[i] = 0;
    if (abs(i - 2 * best_pitch[0]) > 2 && abs(i - 2 * best_pitch[1]) > 2) continue;
    float sum = dot64(x_lp, y + i, len >> 1);
    xcorr[i] = fmaxf(-1, sum);
  }
  find_best_pitch(xcorr, y, len >> 1, max_pitch >> 1, best_pitch);
  int offset = 0;
  if (best_pitch[0] > 0 && best_pitch[0] < (max_pitch >> 1) - 1) {
    float a = xcorr[best_pitch[0] - 1], b = xcorr[best_pitch[0]], c = xcorr[best_pitch[0] + 1];
    if ((c - a) > .7f * (b - a)) offset = 1;
    else if ((a - c) > .7f * (b - c)) offset = -1;
  }
  *pitch = 2 * best_pitch[0] - offset;
}

static float compute_pitch_gain(float xy, float xx, float yy) { return xy / sqrtf(1 + xx * yy); }

static const int second_check[16] = {0, 0, 3, 2, 3, 2, 5, 2, 3, 2, 3, 2, 5, 2, 3, 2};

static float remove_doubling(const float *x, int maxperiod, int minperiod, int N, int *T0_, int prev_period,
                             float prev_gain) {
  int minperiod0 = minperiod;
  maxperiod /= 2; minperiod /= 2; *T0_ /= 2; prev_period /= 2; N /= 2;
  x += maxperiod;
  if (*T0_ >= maxperiod) *T0_ = maxperiod - 1;
  int T, T0;
  T = T0 = *T0_;
  float yy_lookup[(PMAX >> 1) + 1];
  float xx = dot64(x, x, N), xy = dot64(x, x - T0, N);
  yy_lookup[0] = xx;
  {
    float e[PMAX >> 1], S[PMAX >> 1];
    for (int i = 1; i <= maxperiod; ++i) e[i - 1] = x[-i] * x[-i] - x[N - i] * x[N - i];
    if (afo_rnn_eval_order == 1) { /* yy = yy + x[-i]^2 - x[N-i]^2, running */
      float yy_run = xx;
      for (int i = 1; i <= maxperiod; ++i) {
        yy_run = yy_run + x[-i] * x[-i] - x[N - i] * x[N - i];
        yy_lookup[i] = fmaxf(0, yy_run);
      }
    } else {
      scan64(e, S, maxperiod);
      for (int i = 1; i <= maxperiod; ++i) yy_lookup[i] = fmaxf(0, xx + S[i - 1]);
    }
  }
  float yy = yy_lookup[T0];
  float best_xy = xy, best_yy = yy;
  float g, g0;
  g = g0 = compute_pitch_gain(xy, xx, yy);
  for (int k = 2; k <= 15; ++k) {
    int T1 = (2 * T0 + k) / (2 * k);
    if (T1 < minperiod) break;
    int T1b;
    if (k == 2) T1b = (T1 + T0 > maxperiod) ? T0 : T0 + T1;
    else T1b = (2 * second_check[k] * T0 + k) / (2 * k);
    float xy1 = dot64(x, x - T1, N), xy2 = dot64(x, x - T1b, N);
    xy = .5f * (xy1 + xy2);
    yy = .5f * (yy_lookup[T1] + yy_lookup[T1b]);
    float g1 = compute_pitch_gain(xy, xx, yy);
    float cont;
    if (abs(T1 - prev_period) <= 1) cont = prev_gain;
    else if (abs(T1 - prev_period) <= 2 && 5 * k * k < T0) cont = .5f * prev_gain;
    else cont = 0;
    float thresh = fmaxf(.3f, .7f * g0 - cont);
    if (T1 < 3 * minperiod) thresh = fmaxf(.4f, .85f * g0 - cont);
    else if (T1 < 2 * minperiod) thresh = fmaxf(.5f, .9f * g0 - cont);
    if (g1 > thresh) { best_xy = xy; best_yy = yy; T = T1; g = g1; }
  }
  best_xy = fmaxf(0, best_xy);
  float pg = (best_yy <= best_xy) ? 1.0f : best_xy / (best_yy + 1);
  float xc[3];
  for (int k = 0; k < 3; ++k) xc[k] = dot64(x, x - (T + k - 1), N);
  int offset = 0;
  if ((xc[2] - xc[0]) > .7f * (xc[1] - xc[0])) offset = 1;
  else if ((xc[0] - xc[2]) > .7f * (xc[1] - xc[2])) offset = -1;
  if (pg > g) pg = g;
  *T0_ = 2 * T + offset;
  if (*T0_ < minperiod0) *T0_ = minperiod0;
  return pg;
}

/* ------------------------------------------------------------------- RNN
 * Accumulations are explicit fmaf chains in index order (bias, inputs, recurrent inputs): the
 * GPU evaluates the same layers on the f32 matrix cores, whose result is exactly that chain. */
#define WEIGHTS_SCALE (1.f / 256)

static float tansig_approx(float x) {
  if (!(x < 8)) return 1;
  if (!(x > -8)) return -1;
  float sign = 1;
  if (x < 0) { x = -x; sign = -1; }
  int i = (int)floorf(.5f + 25 * x);
  x -= .04f * i;
  float y = tansig_table[i];
  float dy = 1 - y * y;
  y = y + x * dy * (1 - y * x);
  return sign * y;
}
static float sigmoid_approx(float x) { return .5f + .5f * tansig_approx(.5f * x); }

static void dense(const int8_t *w, const int8_t *bias, int n_in, int n_out, int act, float *out, const float *in) {
  /* unit i accumulates bias, then its inputs in ascending j: the loops are interchanged (j outer) so that the units run
   * side by side in vector lanes; the chain per unit is unchanged */
  float sum[96];
  for (int i = 0; i < n_out; ++i) sum[i] = bias[i];
  for (int j = 0; j < n_in; ++j) {
    const float v = in[j];
    const int8_t *wr = w + j * n_out;
    for (int i = 0; i < n_out; ++i) sum[i] = fmaf((float)wr[i], v, sum[i]);
  }
  for (int i = 0; i < n_out; ++i) {
    const float t = WEIGHTS_SCALE * sum[i];
    out[i] = act == 0 ? tansig_approx(t) : sigmoid_approx(t);
  }
}

static void gru(const int8_t *w, const int8_t *u, const int8_t *bias, int M, int N, float *state, const float *in) {
  float z[96], r[96], h[96], sum[96], gated[96];
  const int stride = 3 * N;
  /* per unit: bias, inputs in ascending j, recurrent inputs in ascending j (loops interchanged as in dense()) */
  for (int gate = 0; gate < 3; ++gate) {
    const int off = gate * N;
    for (int i = 0; i < N; ++i) sum[i] = bias[off + i];
    for (int j = 0; j < M; ++j) {
      const float v = in[j];
      const int8_t *wr = w + j * stride + off;
      for (int i = 0; i < N; ++i) sum[i] = fmaf((float)wr[i], v, sum[i]);
    }
    const float *rec = gate == 2 ? gated : state;
    if (gate == 2)
      for (int j = 0; j < N; ++j) gated[j] = state[j] * r[j];
    for (int j = 0; j < N; ++j) {
      const float v = rec[j];
      const int8_t *ur = u + j * stride + off;
      for (int i = 0; i < N; ++i) sum[i] = fmaf((float)ur[i], v, sum[i]);
    }
    if (gate == 0) {
      for (int i = 0; i < N; ++i) z[i] = sigmoid_approx(WEIGHTS_SCALE * sum[i]);
    } else if (gate == 1) {
      for (int i = 0; i < N; ++i) r[i] = sigmoid_approx(WEIGHTS_SCALE * sum[i]);
    } else {
      for (int i = 0; i < N; ++i) {
        float t = WEIGHTS_SCALE * sum[i];
        t = t < 0 ? 0 : t; /* ReLU */
        h[i] = z[i] * state[i] + (1 - z[i]) * t;
      }
    }
  }
  memcpy(state, h, sizeof(float) * N);
}

/* splitmix-style generator -> small int8 weights, identical on every platform */
static uint64_t wrng(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static void fill_i8(int8_t *dst, size_t n, uint64_t *s, int amp) {
  for (size_t i = 0; i < n; ++i) {
    int a = (int)(wrng(s) % (2 * amp + 1)) - amp;
    int b = (int)(wrng(s) % (2 * amp + 1)) - amp;
    dst[i] = (int8_t)((a + b) / 2);
  }
}

void afo_rnn_weights_synthetic(afo_rnn_weights *w, uint64_t seed) {
  uint64_t s = seed;
  fill_i8(w->input_dense_w, sizeof w->input_dense_w, &s, 48);
  fill_i8(w->input_dense_b, sizeof w->input_dense_b, &s, 20);
  fill_i8(w->vad_gru_w, sizeof w->vad_gru_w, &s, 40);
  fill_i8(w->vad_gru_u, sizeof w->vad_gru_u, &s, 40);
  fill_i8(w->vad_gru_b, sizeof w->vad_gru_b, &s, 20);
  fill_i8(w->vad_out_w, sizeof w->vad_out_w, &s, 60);
  fill_i8(w->vad_out_b, sizeof w->vad_out_b, &s, 20);
  fill_i8(w->noise_gru_w, sizeof w->noise_gru_w, &s, 30);
  fill_i8(w->noise_gru_u, sizeof w->noise_gru_u, &s, 30);
  fill_i8(w->noise_gru_b, sizeof w->noise_gru_b, &s, 20);
  fill_i8(w->denoise_gru_w, sizeof w->denoise_gru_w, &s, 24);
  fill_i8(w->denoise_gru_u, sizeof w->denoise_gru_u, &s, 24);
  fill_i8(w->denoise_gru_b, sizeof w->denoise_gru_b, &s, 20);
  fill_i8(w->denoise_out_w, sizeof w->denoise_out_w, &s, 60);
  fill_i8(w->denoise_out_b, sizeof w->denoise_out_b, &s, 40);
}

static void compute_rnn(const afo_rnn_weights *w, afo_rnn_state *st, float *gains, float *vad, const float *feat) {
  float dense_out[24], noise_in[90], denoise_in[114];
  dense(w->input_dense_w, w->input_dense_b, 42, 24, 0, dense_out, feat);
  gru(w->vad_gru_w, w->vad_gru_u, w->vad_gru_b, 24, 24, st->vad_gru_state, dense_out);
  dense(w->vad_out_w, w->vad_out_b, 24, 1, 1, vad, st->vad_gru_state);
  memcpy(noise_in, dense_out, sizeof(float) * 24);
  memcpy(noise_in + 24, st->vad_gru_state, sizeof(float) * 24);
  memcpy(noise_in + 48, feat, sizeof(float) * 42);
  gru(w->noise_gru_w, w->noise_gru_u, w->noise_gru_b, 90, 48, st->noise_gru_state, noise_in);
  memcpy(denoise_in, st->vad_gru_state, sizeof(float) * 24);
  memcpy(denoise_in + 24, st->noise_gru_state, sizeof(float) * 48);
  memcpy(denoise_in + 72, feat, sizeof(float) * 42);
  gru(w->denoise_gru_w, w->denoise_gru_u, w->denoise_gru_b, 114, 96, st->denoise_gru_state, denoise_in);
  dense(w->denoise_out_w, w->denoise_out_b, 96, 22, 1, gains, st->denoise_gru_state);
}

/* debug tap of the last processed frame (tests compare GPU intermediates against it) */
__thread afo_rnn_debug afo_rnn_last; /* per thread: the all-cores baseline leg runs one stream per thread */

/* --------------------------------------------------------- DenoiseState */
void afo_rnn_state_init(afo_rnn_state *st) {
  init_tables();
  memset(st, 0, sizeof(*st));
}

static void biquad_hp(float *y, float *mem, const float *x, int n) {
  const float b0 = -2.0f, b1 = 1.0f, a0 = -1.99599f, a1 = 0.99600f;
  for (int i = 0; i < n; ++i) {
    float xi = x[i];
    float yi = x[i] + mem[0];
    mem[0] = mem[1] + (b0 * xi - a0 * yi);
    mem[1] = (b1 * xi - a1 * yi);
    y[i] = yi;
  }
}

static void pitch_filter(cpx *X, const cpx *P, const float *Ex, const float *Ep, const float *Exp, const float *g) {
  float r[NB], rf[FREQ], newE[NB], norm[NB], normf[FREQ];
  for (int i = 0; i < NB; ++i) {
    if (Exp[i] > g[i]) r[i] = 1;
    else r[i] = Exp[i] * Exp[i] * (1 - g[i] * g[i]) / (.001f + g[i] * g[i] * (1 - Exp[i] * Exp[i]));
    r[i] = sqrtf(fminf(1, fmaxf(0, r[i])));
    r[i] *= sqrtf(Ex[i] / (1e-8f + Ep[i]));
  }
  interp_band_gain(rf, r);
  for (int i = 0; i < FREQ; ++i) {
    X[i].r += rf[i] * P[i].r;
    X[i].i += rf[i] * P[i].i;
  }
  compute_band_energy(newE, X);
  for (int i = 0; i < NB; ++i) norm[i] = sqrtf(Ex[i] / (1e-8f + newE[i]));
  interp_band_gain(normf, norm);
  for (int i = 0; i < FREQ; ++i) {
    X[i].r *= normf[i];
    X[i].i *= normf[i];
  }
}

float afo_rnn_process_frame(const afo_rnn_weights *w, afo_rnn_state *st, float *out, const float *in) {
  float x[FRAME], buf[WINDOW];
  cpx X[FREQ], P[FREQ];
  float Ex[NB], Ep[NB], Exp[NB], features[NFEAT], g[NB], gf[FREQ], Ly[NB], tmp[NB];
  float vad_prob = 0;
  biquad_hp(x, st->mem_hp_x, in, FRAME);
  /* frame_analysis */
  memcpy(buf, st->analysis_mem, sizeof(float) * FRAME);
  memcpy(buf + FRAME, x, sizeof(float) * FRAME);
  memcpy(st->analysis_mem, x, sizeof(float) * FRAME);
  apply_window(buf);
  forward_transform(X, buf);
  compute_band_energy(Ex, X);
  for (int i = 0; i < FREQ; ++i) { afo_rnn_last.X[2 * i] = X[i].r; afo_rnn_last.X[2 * i + 1] = X[i].i; }
  /* pitch */
  memmove(st->pitch_buf, st->pitch_buf + FRAME, sizeof(float) * (PBUF - FRAME));
  memcpy(st->pitch_buf + PBUF - FRAME, x, sizeof(float) * FRAME);
  float pitch_ds[PBUF >> 1];
  pitch_downsample(st->pitch_buf, pitch_ds, PBUF);
  int pitch_index;
  pitch_search(pitch_ds + (PMAX >> 1), pitch_ds, PFRAME, PMAX - 3 * PMIN, &pitch_index);
  pitch_index = PMAX - pitch_index;
  float gain = remove_doubling(pitch_ds, PMAX, PMIN, PFRAME, &pitch_index, st->last_period, st->last_gain);
  st->last_period = pitch_index;
  st->last_gain = gain;
  for (int i = 0; i < WINDOW; ++i) buf[i] = st->pitch_buf[PBUF - WINDOW - pitch_index + i];
  apply_window(buf);
  forward_transform(P, buf);
  for (int i = 0; i < FREQ; ++i) { afo_rnn_last.P[2 * i] = P[i].r; afo_rnn_last.P[2 * i + 1] = P[i].i; }
  compute_band_energy(Ep, P);
  compute_band_corr(Exp, X, P);
  for (int i = 0; i < NB; ++i) Exp[i] = Exp[i] / sqrtf(.001f + Ex[i] * Ep[i]);
  dct(tmp, Exp);
  for (int i = 0; i < NDELTA; ++i) features[NB + 2 * NDELTA + i] = tmp[i];
  features[NB + 2 * NDELTA] -= 1.3f;
  features[NB + 2 * NDELTA + 1] -= 0.9f;
  features[NB + 3 * NDELTA] = .01f * (pitch_index - 300);
  float logMax = -2, follow = -2, E = 0;
  for (int i = 0; i < NB; ++i) {
    Ly[i] = log10f(1e-2f + Ex[i]);
    Ly[i] = fmaxf(logMax - 7, fmaxf(follow - 1.5f, Ly[i]));
    logMax = fmaxf(logMax, Ly[i]);
    follow = fmaxf(follow - 1.5f, Ly[i]);
    E += Ex[i];
  }
  int silence = 0;
  if (E < 0.04f) {
    memset(features, 0, sizeof features);
    silence = 1;
  } else {
    dct(features, Ly);
    features[0] -= 12;
    features[1] -= 4;
    float *ceps_0 = st->cepstral_mem[st->memid];
    float *ceps_1 = (st->memid < 1) ? st->cepstral_mem[CEPS_MEM + st->memid - 1] : st->cepstral_mem[st->memid - 1];
    float *ceps_2 = (st->memid < 2) ? st->cepstral_mem[CEPS_MEM + st->memid - 2] : st->cepstral_mem[st->memid - 2];
    for (int i = 0; i < NB; ++i) ceps_0[i] = features[i];
    st->memid++;
    for (int i = 0; i < NDELTA; ++i) {
      features[i] = ceps_0[i] + ceps_1[i] + ceps_2[i];
      features[NB + i] = ceps_0[i] - ceps_2[i];
      features[NB + NDELTA + i] = ceps_0[i] - 2 * ceps_1[i] + ceps_2[i];
    }
    if (st->memid == CEPS_MEM) st->memid = 0;
    float spec_variability = 0;
    for (int i = 0; i < CEPS_MEM; ++i) {
      float mindist = 1e15f;
      for (int j = 0; j < CEPS_MEM; ++j) {
        float dist = 0;
        for (int k = 0; k < NB; ++k) {
          float t = st->cepstral_mem[i][k] - st->cepstral_mem[j][k];
          dist += t * t;
        }
        if (j != i) mindist = fminf(mindist, dist);
      }
      spec_variability += mindist;
    }
    features[NB + 3 * NDELTA + 1] = spec_variability / CEPS_MEM - 2.1f;
  }
  if (!silence) {
    compute_rnn(w, st, g, &vad_prob, features);
    pitch_filter(X, P, Ex, Ep, Exp, g);
    for (int i = 0; i < NB; ++i) {
      g[i] = fmaxf(g[i], 0.6f * st->lastg[i]);
      st->lastg[i] = g[i];
    }
    interp_band_gain(gf, g);
    for (int i = 0; i < FREQ; ++i) {
      X[i].r *= gf[i];
      X[i].i *= gf[i];
    }
  }
  memcpy(afo_rnn_last.Ex, Ex, sizeof Ex);
  memcpy(afo_rnn_last.Ep, Ep, sizeof Ep);
  memcpy(afo_rnn_last.Exp, Exp, sizeof Exp);
  memcpy(afo_rnn_last.features, features, sizeof features);
  memcpy(afo_rnn_last.gains, g, sizeof g);
  afo_rnn_last.pitch_index = pitch_index;
  afo_rnn_last.pitch_gain = gain;
  afo_rnn_last.silence = silence;
  /* frame_synthesis */
  inverse_transform(buf, X);
  apply_window(buf);
  for (int i = 0; i < FRAME; ++i) out[i] = buf[i] + st->synthesis_mem[i];
  memcpy(st->synthesis_mem, buf + FRAME, sizeof(float) * FRAME);
  return vad_prob;
}

/* ------------------------------------------- wrapper: dsp/rnnoise.rs:45-164 */
void afo_suppressor_init(afo_suppressor *s, float strength, uint64_t weight_seed) {
  memset(s, 0, sizeof(*s));
  afo_rnn_state_init(&s->st);
  afo_rnn_weights_synthetic(&s->w, weight_seed);
  const float sample_rate = 48000.0f, smoothing_ms = 15.0f;
  const float tau = smoothing_ms / 1000.0f;
  const float frame_dt = (float)FRAME / sample_rate;
  s->smoothing_coeff = 1.0f - expf(-(frame_dt / tau)); /* rnnoise.rs:45-51 */
  s->smoothed_strength = 1.0f;
  s->strength = strength < 0 ? 0 : (strength > 1 ? 1 : strength);
}

/* rnnoise.rs:89-111 */
float afo_scale_sample_for_model(float sample) {
  const float PCM_SCALE = 32768.0f, LIMIT = 32760.0f, LIMIT_UNIT = 32760.0f / 32768.0f, THR = 0.98f;
  const float KNEE = 1.0f - 0.98f;
  float v;
  if (!isfinite(sample)) v = 0.0f;
  else {
    float sign = sample > 0 ? 1.0f : (sample < 0 ? -1.0f : (signbit(sample) ? -1.0f : 1.0f));
    float magnitude = fabsf(sample);
    if (magnitude <= THR) v = sample;
    else {
      float over = magnitude - THR;
      float compressed = over / (over + KNEE);
      float softened = THR + (LIMIT_UNIT - THR) * compressed;
      v = sign * fminf(softened, LIMIT_UNIT);
    }
  }
  float scaled = v * PCM_SCALE;
  return scaled < -LIMIT ? -LIMIT : (scaled > LIMIT ? LIMIT : scaled);
}

/* one 480-sample frame of RNNoiseProcessor::process_frames (rnnoise.rs:122-164) */
void afo_suppressor_process_frame(afo_suppressor *s, float *out, const float *dry) {
  float scaled[FRAME], wet[FRAME];
  for (int i = 0; i < FRAME; ++i) scaled[i] = afo_scale_sample_for_model(dry[i]);
  afo_rnn_process_frame(&s->w, &s->st, wet, scaled);
  for (int i = 0; i < FRAME; ++i) wet[i] /= 32768.0f;
  s->smoothed_strength = s->strength * s->smoothing_coeff + s->smoothed_strength * (1.0f - s->smoothing_coeff);
  float strength = s->smoothed_strength;
  if (strength < 1.0f) {
    for (int i = 0; i < FRAME; ++i) wet[i] = (strength * wet[i]) + ((1.0f - strength) * dry[i]);
  }
  memcpy(out, wet, sizeof(float) * FRAME);
}

/* whole clip in complete frames (the tail shorter than a frame stays buffered, like the ring) */
size_t afo_suppressor_process(afo_suppressor *s, float *out, const float *in, size_t n) {
  size_t frames = n / FRAME;
  for (size_t f = 0; f < frames; ++f) afo_suppressor_process_frame(s, out + f * FRAME, in + f * FRAME);
  return frames * FRAME;
}

/* the same with the per-frame pitch decision and silence flag recorded (tests count differing decisions) */
size_t afo_suppressor_process_traced(afo_suppressor *s, float *out, const float *in, size_t n, int32_t *pitch, int32_t *silence) {
  size_t frames = n / FRAME;
  for (size_t f = 0; f < frames; ++f) {
    afo_suppressor_process_frame(s, out + f * FRAME, in + f * FRAME);
    if (pitch) pitch[f] = afo_rnn_last.pitch_index;
    if (silence) silence[f] = afo_rnn_last.silence;
  }
  return frames * FRAME;
}

/* bin/rnnoise_benchmark.rs:51-117 file protocol core: clamp(+-1)*32768 -> process_frame -> /32768 */
void afo_rnnoise_benchmark_frames(const float *in, float *out, size_t n, uint64_t weight_seed) {
  afo_rnn_weights w;
  afo_rnn_state st;
  afo_rnn_weights_synthetic(&w, weight_seed);
  afo_rnn_state_init(&st);
  float frame_in[FRAME], frame_out[FRAME];
  for (size_t start = 0; start < n; start += FRAME) {
    size_t len = n - start < FRAME ? n - start : FRAME;
    memset(frame_in, 0, sizeof frame_in);
    for (size_t i = 0; i < len; ++i) {
      float v = in[start + i];
      v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
      frame_in[i] = v * 32768.0f;
    }
    afo_rnn_process_frame(&w, &st, frame_out, frame_in);
    for (size_t i = 0; i < len; ++i) out[start + i] = frame_out[i] / 32768.0f;
  }
}

/* ---------------------------------------------------------------------------------------------
 * RNNoiseProcessor with its two fixed rings (rnnoise.rs:11,25-42,114-245 over audio/rt.rs:146-248):
 * push_samples -> process_frames -> pop / read, pending_input, set_enabled (bypass moves the input
 * ring to the output ring untouched), soft_reset (= flush_buffers), reset. */
static size_t ring_remaining(const afo_ring *r) { return AFO_RNN_RING_CAPACITY - r->len; }
static void ring_clear(afo_ring *r) { r->head = 0; r->len = 0; }
static size_t ring_push_slice(afo_ring *r, const float *v, size_t n) { /* rt.rs:189-197 */
  size_t written = n < ring_remaining(r) ? n : ring_remaining(r);
  for (size_t k = 0; k < written; ++k) r->data[(r->head + r->len + k) % AFO_RNN_RING_CAPACITY] = v[k];
  r->len += written;
  return written;
}
static size_t ring_pop_into(afo_ring *r, float *out, size_t n) { /* rt.rs:209-221 */
  size_t count = n < r->len ? n : r->len;
  for (size_t k = 0; k < count; ++k) out[k] = r->data[(r->head + k) % AFO_RNN_RING_CAPACITY];
  r->head = (r->head + count) % AFO_RNN_RING_CAPACITY;
  r->len -= count;
  if (r->len == 0) r->head = 0;
  return count;
}
static size_t ring_move_into(afo_ring *from, afo_ring *to) { /* rt.rs:223-235 */
  size_t moved = 0;
  float scratch[64];
  while (from->len > 0 && ring_remaining(to) > 0) {
    size_t count = from->len;
    if (count > ring_remaining(to)) count = ring_remaining(to);
    if (count > 64) count = 64;
    size_t popped = ring_pop_into(from, scratch, count);
    if (popped == 0) break;
    moved += ring_push_slice(to, scratch, popped);
  }
  return moved;
}

void afo_processor_init(afo_rnnoise_processor *p, float strength, uint64_t weight_seed) {
  memset(p, 0, sizeof(*p));
  afo_suppressor_init(&p->core, strength, weight_seed);
  p->enabled = 1; /* rnnoise.rs:58 */
}
void afo_processor_set_strength(afo_rnnoise_processor *p, float v) { /* rnnoise.rs:67-72 */
  p->core.strength = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
}
float afo_processor_get_strength(const afo_rnnoise_processor *p) { return p->core.strength; }
size_t afo_processor_push_samples(afo_rnnoise_processor *p, const float *v, size_t n) { return ring_push_slice(&p->input, v, n); }
void afo_processor_process_frames(afo_rnnoise_processor *p) { /* rnnoise.rs:122-164 */
  if (!p->enabled) {
    ring_move_into(&p->input, &p->output);
    return;
  }
  float dry[FRAME], out[FRAME];
  while (p->input.len >= FRAME && ring_remaining(&p->output) >= FRAME) {
    if (ring_pop_into(&p->input, dry, FRAME) != FRAME) break;
    afo_suppressor_process_frame(&p->core, out, dry);
    ring_push_slice(&p->output, out, FRAME);
  }
}
size_t afo_processor_available_samples(const afo_rnnoise_processor *p) { return p->output.len; }
size_t afo_processor_pending_input(const afo_rnnoise_processor *p) { return p->input.len; }
size_t afo_processor_read_samples(afo_rnnoise_processor *p, float *out, size_t n) { /* rnnoise.rs:185-188 */
  size_t count = n < p->output.len ? n : p->output.len;
  return ring_pop_into(&p->output, out, count);
}
size_t afo_processor_drain_pending_input(afo_rnnoise_processor *p, float *out, size_t cap) { /* rnnoise.rs:240-244 */
  size_t count = cap < p->input.len ? cap : p->input.len;
  return ring_pop_into(&p->input, out, count);
}
void afo_processor_set_enabled(afo_rnnoise_processor *p, int on) { p->enabled = on != 0; }
void afo_processor_soft_reset(afo_rnnoise_processor *p) { /* rnnoise.rs:216-232 */
  ring_clear(&p->input);
  ring_clear(&p->output);
}
void afo_processor_reset(afo_rnnoise_processor *p) { /* rnnoise.rs:205-210: DenoiseState::new(), smoothing state kept */
  afo_rnn_state_init(&p->core.st);
  ring_clear(&p->input);
  ring_clear(&p->output);
}
