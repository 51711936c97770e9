/*
 * audioforge_mi.h -- C ABI of the MI355X-native batched voice-DSP engine.
 *
 * The reference (FueledByRedBull/audio-forge, rust-core) exposes its per-frame voice
 * chain to Python through PyO3 (`mic_eq.mic_eq_core`, rust-core/src/lib.rs:301-350);
 * it has no C ABI of its own.  This header is the boundary a maintainer binds instead
 * (ctypes / cffi / pyo3-ffi): an `af_engine` is N independent copies of the reference's
 * `OfflineDspBlockProcessor` (rust-core/src/audio/processor/block_processor.rs:31-60),
 * one per 48 kHz mono stream, resident on one GPU, driven through the same setter
 * surface the reference's structs have.  Every entry point cites the reference method
 * it replaces.  All functions return 0 (AF_OK) or a negative af_status; the message of
 * the last failure on the calling thread is af_last_error().
 *
 * Threading: an engine is not thread safe; use one host thread per engine/device.
 * Ownership: every buffer is caller-owned; the engine copies what it keeps.
 */
#ifndef AUDIOFORGE_MI_H
#define AUDIOFORGE_MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum af_status {
  AF_OK = 0,
  AF_ERR_INVALID_ARGUMENT = -1, /* PyValueError in the reference binding            */
  AF_ERR_BACKEND = -2,          /* PyRuntimeError: HIP failure / no device          */
  AF_ERR_NON_FINITE = -3,       /* "audio must contain only finite samples"         */
  AF_ERR_STATE = -4,            /* setter used after streaming started (see below)  */
  AF_ERR_UNSUPPORTED = -5
} af_status;

typedef struct af_engine af_engine;

/* ---- stable public filter ids: rust-core/src/dsp/eq.rs:44-53 ---- */
enum { AF_EQ_LOW_SHELF = 0, AF_EQ_BELL = 1, AF_EQ_HIGH_SHELF = 2, AF_EQ_NOTCH = 3,
       AF_EQ_HIGH_PASS = 4, AF_EQ_LOW_PASS = 5 };

/* EqBandConfig, rust-core/src/dsp/eq.rs:112-120 */
typedef struct af_eq_band_config {
  int32_t filter_type;
  double frequency_hz;
  double gain_db;
  double q;
  int32_t slope_db_per_octave;
  int32_t enabled;
} af_eq_band_config;

/* OfflineDspBlockStats (block_processor.rs:1-28) + the two per-block energy sums that
 * simulate_auto_eq_chain accumulates around each block (python_api.rs:515-553). */
typedef struct af_block_stats {
  float input_sample_peak;
  float output_sample_peak;
  float true_peak_limiter_input_peak;
  float output_true_peak;
  float limiter_peak_gain_reduction_db;
  float true_peak_limiter_gain_reduction_db;
  float compressor_gain_reduction_db;
  float deesser_gain_reduction_db;
  double input_square_sum;
  double output_square_sum;
  uint32_t true_peak_limited_events; /* 0 or 1 per block, true_peak.rs:376 */
  uint32_t non_finite_output;
  /* Compressor::current_makeup_gain / auto_makeup_activity / auto_makeup_activity_reliability at the end of
   * the block (compressor.rs:346-363) -- the traces of simulate_auto_makeup_control */
  float compressor_makeup_gain_db;
  float auto_makeup_activity;
  float auto_makeup_reliability;
  float reserved;
} af_block_stats;

/* layout of the audio buffers handed to af_engine_process_* */
enum { AF_LAYOUT_STREAM_MAJOR = 0 /* [stream][time] */, AF_LAYOUT_TIME_MAJOR = 1 /* [time][stream] */ };

/* kernel variants (all produce the same samples) */
enum { AF_KERNEL_AUTO = 0, AF_KERNEL_LANE_PER_STREAM = 1, AF_KERNEL_PHASED = 2, AF_KERNEL_QUAD = 3, AF_KERNEL_STAGED = 4,
       AF_KERNEL_ROLES = 5 /* wave roles inside one workgroup per 64 streams, LDS hand-over (csrc/af_roles.hip) */ };

int af_version(void);
const char *af_last_error(void);
/* number of HIP devices visible; negative status when the runtime is unusable */
int af_device_count(void);

/* ---- lifecycle ------------------------------------------------------------------ */
/* OfflineDspBlockProcessor::new(sample_rate), block_processor.rs:46-60, for
 * `n_streams` independent streams on HIP device `device`.  No GPU work happens until
 * the first af_engine_process_* call (configuration is pure host work). */
int af_engine_create(double sample_rate, int32_t n_streams, int32_t device, af_engine **out);
void af_engine_destroy(af_engine *engine);
/* Re-arm every stream with the configured initial state (a fresh processor with the
 * same setter history); afterwards setters are accepted again. */
int af_engine_reset(af_engine *engine);
int32_t af_engine_n_streams(const af_engine *engine);

/* Setters mirror the reference structs and may be called until the first
 * af_engine_process_* call; afterwards they return AF_ERR_STATE (live retuning of
 * resident streams is out of scope -- the reference does it through its realtime
 * control plane, audio/processor/control.rs). */

/* ---- chain switches: block_processor.rs:62-84 ----------------------------------- */
int af_engine_set_deesser_enabled(af_engine *e, int32_t enabled);
int af_engine_set_eq_enabled(af_engine *e, int32_t enabled);
int af_engine_set_compressor_enabled(af_engine *e, int32_t enabled);
int af_engine_set_limiter_enabled(af_engine *e, int32_t enabled);
int af_engine_set_eq_before_deesser(af_engine *e, int32_t enabled);
/* Samples per reference block: python_api.rs:512-514 uses round(0.020*fs)=960,
 * the golden test 480 (tests.rs:1825).  Each af_engine_process_* call is cut into
 * blocks of this size (last one short), exactly like `audio.chunks(n)`. */
int af_engine_set_control_block_samples(af_engine *e, int32_t samples);
/* python_api.rs:517-520: non-finite input samples become 0 (on by default) */
int af_engine_set_input_scrub_enabled(af_engine *e, int32_t enabled);

/* ---- realtime front end (off by default: the offline simulator has none) -------- */
/* sanitize_and_clamp_input_inplace, audio/processor/routing.rs:802-823 */
int af_engine_set_input_clamp_enabled(af_engine *e, int32_t enabled);
/* apply_input_pre_filter: DC block + 80 Hz high-pass, routing.rs:826-843 */
int af_engine_set_prefilter_enabled(af_engine *e, int32_t enabled, int32_t apply_fixed_highpass);

/* ---- RNNoise suppressor: rust-core/src/dsp/rnnoise.rs (RNNoiseProcessor) -------------------------
 * Runs between the front end and the EQ (dsp_loop.rs:1521-1599).  48 kHz only.  The suppressor eats whole
 * 480-sample frames: what a process call leaves over waits in the engine for the next call (rnnoise.rs:114-164), and a
 * call returns floor((pending + n) / 480) * 480 samples per stream (see af_engine_stream_host below).  Output is delayed by one frame
 * (latency_samples() = 480, rnnoise.rs:313-315).  The network weights of nnnoiseless 0.5.2 are not
 * available offline: engines start on seeded synthetic weights in the real layout; load the real
 * ones with af_suppressor_load_weights (blob = the model's fifteen int8 arrays: input_dense w,b;
 * vad_gru w,u,b; vad_output w,b; noise_gru w,u,b; denoise_gru w,u,b; denoise_output w,b). */
int af_engine_set_suppressor_enabled(af_engine *e, int32_t enabled);
int af_engine_set_suppressor_strength(af_engine *e, float strength);  /* rnnoise.rs:67-72, live */
int af_suppressor_set_synthetic_weights(af_engine *e, uint64_t seed);
int af_suppressor_load_weights(af_engine *e, const int8_t *blob, size_t bytes);
/* 1: the file protocol of bin/rnnoise_benchmark.rs:51-117 (clamp(+-1)*32768 in, /32768 out, no mix) */
int af_suppressor_set_raw_protocol(af_engine *e, int32_t enabled);
int32_t af_suppressor_latency_samples(const af_engine *e);
/* test tap: RNNoiseProcessor::scale_sample_for_model (rnnoise.rs:89-111; pinned by rnnoise.rs:335-352) evaluated
 * element-wise by the device function the pre-pass kernel uses; host pointers */
int af_suppressor_debug_scale_for_model(const float *in, float *out, int64_t n, int32_t device);
/* test tap: (silence flag, pitch index) of every frame of the last process call, [frame][stream][2] int32 */
int af_suppressor_set_trace_enabled(af_engine *e, int32_t enabled);
int64_t af_suppressor_trace_frames(const af_engine *e);
int af_suppressor_read_trace(af_engine *e, int32_t *out, int64_t capacity_frames);
/* test tap: the analysis record (158 floats/ints) and spectra X, P (481 complex each) of one
 * (frame, stream) cell of the last suppressor window */
int af_suppressor_debug_read(af_engine *e, int32_t frame, int32_t stream, float *record, float *x_spectrum,
                             float *p_spectrum);

/* ---- ParametricEQ: rust-core/src/dsp/eq.rs --------------------------------------- */
int af_eq_set_band_frequency(af_engine *e, int32_t band, double frequency_hz); /* eq.rs:419-425 */
int af_eq_set_band_gain(af_engine *e, int32_t band, double gain_db);           /* eq.rs:406-412 */
int af_eq_set_band_q(af_engine *e, int32_t band, double q);                    /* eq.rs:432-438 */
int af_eq_set_band_config(af_engine *e, int32_t band, const af_eq_band_config *c); /* eq.rs:468-472 */
int af_eq_reset(af_engine *e);                                                 /* eq.rs:395-399 */
/* EqBandConfig::validate, eq.rs:140-201; message via af_last_error() */
int af_eq_band_config_validate(const af_eq_band_config *c, int32_t index, double sample_rate);

/* ---- Compressor: rust-core/src/dsp/compressor.rs:210-371 ------------------------ */
int af_compressor_set_threshold(af_engine *e, double threshold_db);
int af_compressor_set_ratio(af_engine *e, double ratio);
int af_compressor_set_attack_time(af_engine *e, double attack_ms);
int af_compressor_set_release_time(af_engine *e, double release_ms);
int af_compressor_set_makeup_gain(af_engine *e, double makeup_gain_db);
int af_compressor_set_adaptive_release(af_engine *e, int32_t enabled);
int af_compressor_set_base_release_time(af_engine *e, double release_ms);
int af_compressor_set_auto_makeup_enabled(af_engine *e, int32_t enabled);
int af_compressor_set_target_lufs(af_engine *e, double target_lufs);
int af_compressor_set_sidechain_highpass_enabled(af_engine *e, int32_t enabled);
int af_compressor_set_noise_reference_reliability(af_engine *e, double reliability); /* compressor.rs:351-353 */
/* AutoMakeupActivityInput (compressor.rs:32-37) for the NEXT af_engine_process_* call: one speech
 * posterior per control block (shared by all streams, or [block][stream] when per_stream != 0) plus
 * the three scalars simulate_auto_makeup_control passes (python_api.rs:211-219).  n_blocks = 0
 * clears the evidence (process_block_inplace without evidence, compressor.rs:695-697).  May be
 * called between process calls. */
int af_compressor_set_activity_evidence(af_engine *e, const double *vad_probabilities, int64_t n_blocks,
                                        int32_t per_stream, double vad_reliability, double noise_floor_db,
                                        double live_noise_reliability);

/* ---- Limiter: rust-core/src/dsp/limiter.rs:139-184 ------------------------------ */
int af_limiter_set_ceiling(af_engine *e, double ceiling_db);
int af_limiter_set_release_time(af_engine *e, double release_ms);
int af_limiter_set_lookahead_ms(af_engine *e, double lookahead_ms);
double af_limiter_ceiling_db(const af_engine *e);
int32_t af_limiter_lookahead_samples(const af_engine *e);

/* ---- TruePeakLimiter: rust-core/src/dsp/true_peak.rs:304-313 --------------------- */
int af_true_peak_limiter_set_release_ms(af_engine *e, float release_ms);

/* ---- DeEsser: rust-core/src/dsp/deesser.rs:288-353 ------------------------------ */
int af_deesser_set_auto_enabled(af_engine *e, int32_t enabled);
int af_deesser_set_auto_amount(af_engine *e, double amount);
int af_deesser_set_low_cut_hz(af_engine *e, double hz);
int af_deesser_set_high_cut_hz(af_engine *e, double hz);
int af_deesser_set_threshold_db(af_engine *e, double db);
int af_deesser_set_ratio(af_engine *e, double ratio);
int af_deesser_set_attack_ms(af_engine *e, double ms);
int af_deesser_set_release_ms(af_engine *e, double ms);
int af_deesser_set_max_reduction_db(af_engine *e, double db);

/* ---- processing ------------------------------------------------------------------ */
/* OfflineDspBlockProcessor::process_block_with_stats (block_processor.rs:106-161)
 * applied to every stream, for ceil(n/control_block) consecutive blocks.
 *
 * _device: `in`/`out` are device pointers on the engine's device; element (s, t) is at
 *   in[s*stream_stride + t] (stream-major) or in[t*stream_stride + s] (time-major);
 *   `hip_stream` is a hipStream_t (NULL = default stream).  Asynchronous: returns
 *   after enqueueing.  `in` may equal `out`.
 * _host: host pointers; copies in, runs, copies out, synchronises. */
int af_engine_process_device(af_engine *e, const float *in, float *out, int64_t n_samples,
                             int64_t stream_stride, int32_t layout, void *hip_stream);
int af_engine_process_host(af_engine *e, const float *in, float *out, int64_t n_samples,
                           int32_t layout);
int af_engine_synchronize(af_engine *e);
/* One wake-up of the realtime loop with the suppressor on (dsp_loop.rs:1521-1599: push_samples -> process_frames ->
 * pop_samples_into -> downstream chain): n_in new samples per stream go in ([stream][n_in]), the whole 480-sample frames
 * that are complete come out (*n_out = floor((pending + n_in) / 480) * 480 per stream at out_stride, possibly 0 or more
 * than n_in), the remainder waits in the engine.  af_engine_process_* follow the same rule (af_engine_process_device
 * needs stream_stride >= that count; af_engine_process_host fails when it exceeds n_samples).  Without the suppressor
 * *n_out == n_in.  Reference test: rnnoise.rs:355-370 (400 in -> 0 out, 400 pending; +100 -> 480 out, 20 pending). */
int af_engine_stream_host(af_engine *e, const float *in, int64_t n_in, float *out, int64_t out_stride, int64_t *n_out);
int64_t af_engine_pending_input(const af_engine *e);        /* RNNoiseProcessor::pending_input, rnnoise.rs:234-237 */
int64_t af_engine_last_output_samples(const af_engine *e);  /* samples per stream the last process call produced */
/* Blocks produced by the last process call and their stats, row-major [block][stream]
 * (synchronises).  `capacity` is in rows of af_block_stats. */
int64_t af_engine_last_block_count(const af_engine *e);
int af_engine_read_block_stats(af_engine *e, af_block_stats *out, int64_t capacity);
/* total samples per stream processed since create/reset */
int64_t af_engine_samples_processed(const af_engine *e);
/* ---- presets ---------------------------------------------------------------------
 * The reference configures one processor per stream (python/mic_eq/config_parts/settings.py:543-593).  An engine runs its
 * streams in groups of 64 (one chain workgroup each) and every group may run its own preset: add presets (each starts as a
 * fresh OfflineDspBlockProcessor::new, block_processor.rs:46-60), select the one the chain / EQ / compressor / limiter /
 * de-esser setters address, and map groups to presets (one index per 64 streams).  The control block, the front-end
 * switches, the suppressor and the kernel choice stay engine-wide.  Multi-preset engines run the plain token-ring form:
 * AF_ERR_UNSUPPORTED when a preset enables the de-esser or auto-makeup, or its limiter lookahead does not fit the ring. */
int af_engine_set_preset_count(af_engine *e, int32_t n);
int32_t af_engine_preset_count(const af_engine *e);  /* VALUE */
int af_engine_select_preset(af_engine *e, int32_t preset);
int af_engine_assign_presets(af_engine *e, const int32_t *preset_of_group, int32_t n_groups);
/* choose the kernel variant (AF_KERNEL_*); default AF_KERNEL_AUTO: up to 3072 streams (2048 behind the suppressor) AF_KERNEL_STAGED
 * (the chain as a pipeline of stage kernels, one per recurrence; not built for the de-esser, the front end without the
 * suppressor, more than 16 EQ sections, presets that differ in which stages run, time-major audio: those fall through); else AF_KERNEL_PHASED
 * (the token ring, 64 streams per workgroup) wherever its LDS layout fits, AF_KERNEL_QUAD (16 streams per workgroup) for
 * longer limiter lookaheads, AF_KERNEL_LANE_PER_STREAM otherwise.  The choice is made at the first process call after a
 * reset and kept (kernel 4 keeps its delay lines in buffers of its own).  All variants give the same bits. */
int af_engine_set_kernel(af_engine *e, int32_t kernel);
/* AF_KERNEL_* the most recent chain launch used; VALUE, not a status */
int af_engine_last_kernel(const af_engine *e);
/* tuning of the token-ring kernel: wavefronts per 64-stream group and samples per chunk
 * (built: 16x4, 16x2, 12x4, 12x2, 8x4, 8x2; 0,0 = default) */
int af_engine_set_ring_variant(af_engine *e, int32_t waves, int32_t chunk);
/* HIP-event timing of the kernels launched by the last process call, in milliseconds,
 * measured on the stream the kernels ran on (0 when timing is disabled) */
int af_engine_set_timing_enabled(af_engine *e, int32_t enabled);
int af_engine_last_kernel_ms(af_engine *e, double *ms, int32_t *launches);
/* the same split at the suppressor | chain boundary (front-end pre-pass counts as suppressor time) */
int af_engine_last_stage_ms(af_engine *e, double *suppressor_ms, double *chain_ms);
/* chain launches of the last call: their summed duration and the number of segments (`tail_ms` is always 0: it belonged
 * to a two-launch form of the chain that was measured slower and removed) */
int af_engine_last_chain_launch_ms(af_engine *e, double *first_ms, double *tail_ms, int32_t *segments);

/* ---- product resampler ------------------------------------------------------------------
 * `build_sinc_resampler_with_quality` + `simulate_product_resampler`
 * (rust-core/src/audio/processor/resampling.rs:140-156, 179-261): rubato's asynchronous windowed-sinc
 * resampler (sinc_len taps, 256 oversampled rows, cubic interpolation), driven in chunks of `chunk_size`
 * frames: full chunks, one zero-padded partial chunk, then silent flush chunks until expected + delay
 * frames exist.  One af_resampler serves any number of equally long f64 streams per call; audio is
 * stream-major ([stream][frame], strides in frames).  Windows: resampler_window_from_name,
 * resampling.rs:158-168. */
typedef struct af_resampler af_resampler;
enum { AF_WINDOW_BLACKMAN_HARRIS = 0, AF_WINDOW_BLACKMAN_HARRIS_SQUARED = 1, AF_WINDOW_BLACKMAN = 2,
       AF_WINDOW_BLACKMAN_SQUARED = 3, AF_WINDOW_HANN = 4, AF_WINDOW_HANN_SQUARED = 5 };
/* rubato::calculate_cutoff(sinc_len, window) (resampling.rs:149) */
int af_resampler_calculate_cutoff(int32_t sinc_len, int32_t window, float *out);
/* argument contract of resampling.rs:187-214 (AF_ERR_INVALID_ARGUMENT with the reference's messages);
 * AF_ERR_UNSUPPORTED when sinc_len / ratio need a longer input span than the kernel's LDS tile */
int af_resampler_create(uint32_t input_rate, uint32_t output_rate, int64_t chunk_size, int32_t sinc_len,
                        int32_t window, int32_t device, af_resampler **out);
void af_resampler_destroy(af_resampler *r);
/* Resampler::output_delay (resampling.rs:216); VALUE, not a status */
int af_resampler_output_delay(const af_resampler *r);
/* round(n_in * output_rate / input_rate) (resampling.rs:217-218); VALUE */
int64_t af_resampler_expected_frames(const af_resampler *r, int64_t n_in);
/* effective sinc length (rounded up to a multiple of 8 like the crate); VALUE */
int af_resampler_sinc_len(const af_resampler *r);
/* the 256 x sinc_len coefficient table, row-major (diagnostics / cross-checks) */
int af_resampler_copy_sinc_table(const af_resampler *r, double *out);
/* host only: frames the reference's driver loop returns for n_in input frames, and how many chunks it runs */
int af_resampler_plan(af_resampler *r, int64_t n_in, int64_t *n_out, int64_t *blocks);
/* all streams, all chunks, one launch; writes af_resampler_plan's n_out frames per stream */
int af_resampler_process_device(af_resampler *r, const double *d_in, double *d_out, int64_t n_in,
                                int32_t n_streams, int64_t in_stride, int64_t out_stride, void *hip_stream);
/* host buffers (checks "samples must be finite" -> AF_ERR_NON_FINITE) */
int af_resampler_process_host(af_resampler *r, const double *in, double *out, int64_t n_in, int32_t n_streams,
                              int64_t in_stride, int64_t out_stride);
/* HIP-event time of the last launch */
int af_resampler_last_kernel_ms(af_resampler *r, double *ms);

/* ---- noise gate -------------------------------------------------------------------------------
 * NoiseGate (rust-core/src/dsp/gate.rs) on the path `simulate_gate_suppressor_order` exercises
 * (python_api.rs:312-319 builds the gate without a VadAutoGate, so process_block_inplace runs the per-sample
 * downward expander, gate.rs:626-637).  One-shot over whole clips from the initial state (gate.rs:158-225);
 * `vad_mode` != 0 = GateMode::VadAssisted/VadOnly (arms the chatter auto-relax, gate.rs:598-601).
 * gain_trace: [ceil(n / trace_block)][n_streams] `current_gain()` at the end of every block (python_api.rs:356);
 * chatter_events: [n_streams] `chatter_event_count()`.  Either may be null. */
int af_gate_process_host(const float *in, float *out, int64_t n_samples, int32_t n_streams, int64_t stream_stride,
                         double threshold_db, double attack_ms, double release_ms, double sample_rate,
                         int32_t vad_mode, int32_t trace_block, float *gain_trace, uint64_t *chatter_events,
                         int32_t device);

/* ---- integrated loudness ------------------------------------------------------------------
 * measure_integrated_loudness (rust-core/src/lib.rs:290-298 over dsp/loudness.rs:43-83): BS.1770 gated loudness of
 * whole clips, ebur128 `Mode::I | Mode::HISTOGRAM`, mono.  One value per stream; `status` (optional, [n_streams])
 * holds AF_OK, AF_ERR_NON_FINITE ("samples must be finite") or AF_ERR_UNSUPPORTED (nothing passed the gates:
 * "audio did not produce a finite gated loudness"); the call returns the first failure.  Sample rates:
 * loudness.rs:36-41. */
int af_measure_integrated_loudness_device(const float *d_audio, int64_t n_samples, int32_t n_streams,
                                          int64_t stream_stride, uint32_t sample_rate, int32_t device,
                                          double *lufs, int32_t *status);
int af_measure_integrated_loudness_host(const float *audio, int64_t n_samples, int32_t n_streams,
                                        int64_t stream_stride, uint32_t sample_rate, int32_t device,
                                        double *lufs, int32_t *status);

/* ---- NoiseSuppressor: rust-core/src/dsp/noise_suppressor.rs:18-194 ----------------------------------------
 * The runtime-selected suppressor interface (`NoiseModel`, trait `NoiseSuppressor`, `new_noise_suppression_engine`) for a
 * batch of streams that advance in lock step: sample counts are per stream, audio is [stream][stride] host memory, the
 * two fixed rings hold 8192 + 480 samples per stream (rnnoise.rs:11).  Model ids as NoiseModel::from_id / id(); the
 * DeepFilterNet variants parse but cannot be created (AF_ERR_UNSUPPORTED: the reference loads them from a runtime
 * library + model archives that are not in its checkout, deepfilter_ffi.rs:9-16), and af_noise_model_available lists
 * what a default build of the reference lists: RNNoise. */
enum { AF_NOISE_MODEL_RNNOISE = 0, AF_NOISE_MODEL_DEEPFILTER_LL = 1, AF_NOISE_MODEL_DEEPFILTER = 2 };
typedef struct af_noise_suppressor af_noise_suppressor;
int af_noise_model_from_id(const char *id, int32_t *model);           /* noise_suppressor.rs:58-67 */
const char *af_noise_model_id(int32_t model);                         /* noise_suppressor.rs:47-55; VALUE */
const char *af_noise_model_display_name(int32_t model);               /* noise_suppressor.rs:36-44; VALUE */
int32_t af_noise_model_available(int32_t *models, int32_t capacity);  /* noise_suppressor.rs:70-84; VALUE: count */
int af_noise_suppressor_create(int32_t model, int32_t n_streams, int32_t device, af_noise_suppressor **out); /* :168-194 */
void af_noise_suppressor_destroy(af_noise_suppressor *s);
/* the engine behind it (weights: af_suppressor_load_weights / af_suppressor_set_synthetic_weights before the first frame) */
af_engine *af_noise_suppressor_engine(af_noise_suppressor *s);
/* VALUE functions: sample counts per stream (negative af_status on a bad argument) */
int64_t af_noise_suppressor_push_samples(af_noise_suppressor *s, const float *samples, int64_t n, int64_t stride);
int af_noise_suppressor_process_frames(af_noise_suppressor *s);
int64_t af_noise_suppressor_available_samples(const af_noise_suppressor *s);
int64_t af_noise_suppressor_pending_input(const af_noise_suppressor *s);
int64_t af_noise_suppressor_pop_samples_into(af_noise_suppressor *s, float *out, int64_t count, int64_t stride);
int64_t af_noise_suppressor_drain_pending_input(af_noise_suppressor *s, float *out, int64_t capacity, int64_t stride);
int af_noise_suppressor_set_strength(af_noise_suppressor *s, float value);   /* clamps to [0, 1] */
float af_noise_suppressor_get_strength(const af_noise_suppressor *s);        /* VALUE */
int af_noise_suppressor_set_enabled(af_noise_suppressor *s, int32_t enabled); /* disabled = bit-exact passthrough */
int32_t af_noise_suppressor_is_enabled(const af_noise_suppressor *s);        /* VALUE */
int af_noise_suppressor_soft_reset(af_noise_suppressor *s);                  /* rings cleared, model state kept */
int af_noise_suppressor_reset(af_noise_suppressor *s);                       /* rnnoise.rs:205-210 */
int32_t af_noise_suppressor_model_type(const af_noise_suppressor *s);        /* VALUE */
int32_t af_noise_suppressor_latency_samples(const af_noise_suppressor *s);   /* VALUE: 480 */
int32_t af_noise_suppressor_backend_available(const af_noise_suppressor *s); /* VALUE */
int32_t af_noise_suppressor_backend_failed(const af_noise_suppressor *s);    /* VALUE */
const char *af_noise_suppressor_backend_error(const af_noise_suppressor *s); /* VALUE: NULL = none */

/* ---- stateless helpers ----------------------------------------------------------- */
/* eq_magnitude_response, lib.rs:99-150 (legacy (freq, gain_db, q) x 10 bands) */
int af_eq_magnitude_response(const double *frequencies_hz, size_t n, const double bands[10][3],
                             double sample_rate, double *out_db);
/* eq_magnitude_response_v2, lib.rs:191-212 */
int af_eq_magnitude_response_v2(const double *frequencies_hz, size_t n,
                                const af_eq_band_config bands[10], double sample_rate,
                                double *out_db);
/* engine's configured EQ, target response: ParametricEQ::magnitude_response_db, eq.rs:511-527 */
int af_engine_eq_magnitude_response(const af_engine *e, const double *frequencies_hz, size_t n,
                                    double *out_db);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOFORGE_MI_H */
