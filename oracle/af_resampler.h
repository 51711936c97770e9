/* af_resampler.h -- CPU restatement of the product resampler (TEST INFRASTRUCTURE ONLY).
 *
 * The reference resamples with the third-party crate rubato 0.14.1 (Cargo.lock:1051-1060), whose source is
 * not in the container: `SincFixedIn::<f64>::new(ratio, 1.2, {sinc_len 128, f_cutoff =
 * calculate_cutoff(128, Blackman), Cubic, oversampling 256, Blackman}, chunk 1024, 1 channel)`
 * (rust-core/src/audio/processor/resampling.rs:140-156) driven by `simulate_product_resampler`
 * (resampling.rs:179-261).  This file restates the crate's published algorithm (asynchronous
 * windowed-sinc interpolation, 256 oversampled sinc rows, cubic interpolation between four neighbouring
 * rows, chunked processing with a 2*sinc_len history) and the reference's driver loop.
 *
 * PARITY: sample parity with the crate is UNPINNED (the crate picks an AVX/SSE/scalar dot product at
 * run time; summation order is not defined).  What IS pinned, in tests/test_oracle_resampler.py, are the
 * reference's own published measurements of this configuration (evaluation/resampler-quality-report.json:
 * block counts, impulse locations, delays, round-trip SNR, alias/image rejection, pass-band error), which
 * this restatement reproduces -- see DESIGN.md for the digits.
 */
#ifndef AF_RESAMPLER_H
#define AF_RESAMPLER_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { AFO_WIN_BLACKMAN_HARRIS = 0, AFO_WIN_BLACKMAN_HARRIS2, AFO_WIN_BLACKMAN, AFO_WIN_BLACKMAN2, AFO_WIN_HANN, AFO_WIN_HANN2 };

typedef struct afo_resampler afo_resampler;

/* rubato::calculate_cutoff(sinc_len, window) as f32 */
float afo_resampler_calculate_cutoff(size_t sinc_len, int window);
/* f_cutoff <= 0 selects calculate_cutoff */
afo_resampler *afo_resampler_new(uint32_t input_rate, uint32_t output_rate, size_t chunk_size, size_t sinc_len,
                                 int window, float f_cutoff);
void afo_resampler_free(afo_resampler *r);
size_t afo_resampler_output_delay(const afo_resampler *r);
size_t afo_resampler_output_frames_max(const afo_resampler *r);
/* one chunk of exactly chunk_size input frames (NULL = a chunk of zeros); returns the frames produced */
size_t afo_resampler_process_chunk(afo_resampler *r, const double *in, double *out);
/* the 256 x sinc_len coefficient table, row-major (for the GPU plan's cross-check) */
const double *afo_resampler_sinc_table(const afo_resampler *r);

/* simulate_product_resampler (resampling.rs:179-261): returns the number of frames written to `out`
 * (< 0: out_capacity too small, -needed), and the delay / expected frame count / number of timed blocks. */
int64_t afo_simulate_product_resampler(const double *samples, size_t n, uint32_t input_rate, uint32_t output_rate,
                                       size_t chunk_size, size_t sinc_len, int window, float f_cutoff, double *out,
                                       size_t out_capacity, size_t *delay, size_t *expected_frames, size_t *blocks);

#ifdef __cplusplus
}
#endif
#endif
