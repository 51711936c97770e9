/* af_rnnoise.h -- RNNoise suppressor restatement (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED,
 * see the header of af_rnnoise.c). */
#ifndef AF_RNNOISE_ORACLE_H
#define AF_RNNOISE_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AFO_RNN_FRAME 480
#define AFO_RNN_BANDS 22
#define AFO_RNN_FEATURES 42

/* int8 weights in the layout of the public RNNoise model: dense [in][out], GRU [in][3*units] (z|r|h) */
typedef struct {
  int8_t input_dense_w[42 * 24], input_dense_b[24];
  int8_t vad_gru_w[24 * 72], vad_gru_u[24 * 72], vad_gru_b[72];
  int8_t vad_out_w[24 * 1], vad_out_b[1];
  int8_t noise_gru_w[90 * 144], noise_gru_u[48 * 144], noise_gru_b[144];
  int8_t denoise_gru_w[114 * 288], denoise_gru_u[96 * 288], denoise_gru_b[288];
  int8_t denoise_out_w[96 * 22], denoise_out_b[22];
} afo_rnn_weights;

typedef struct {
  float analysis_mem[480];
  float cepstral_mem[8][22];
  int memid;
  float synthesis_mem[480];
  float pitch_buf[1728];
  float last_gain;
  int last_period;
  float mem_hp_x[2];
  float lastg[22];
  float vad_gru_state[24], noise_gru_state[48], denoise_gru_state[96];
} afo_rnn_state;

typedef struct {
  float Ex[22], Ep[22], Exp[22], features[42], gains[22];
  float X[962], P[962];
  int pitch_index, silence;
  float pitch_gain;
} afo_rnn_debug;
extern __thread afo_rnn_debug afo_rnn_last;

void afo_rnn_weights_synthetic(afo_rnn_weights *w, uint64_t seed);
void afo_rnn_state_init(afo_rnn_state *st);
/* DenoiseState::process_frame: 480 samples in the +-32768 range; returns the VAD probability */
float afo_rnn_process_frame(const afo_rnn_weights *w, afo_rnn_state *st, float *out, const float *in);

/* RNNoiseProcessor (rust-core/src/dsp/rnnoise.rs:25-164) */
typedef struct {
  afo_rnn_weights w;
  afo_rnn_state st;
  float strength, smoothed_strength, smoothing_coeff;
} afo_suppressor;
void afo_suppressor_init(afo_suppressor *s, float strength, uint64_t weight_seed);
void afo_suppressor_process_frame(afo_suppressor *s, float *out, const float *dry);
size_t afo_suppressor_process(afo_suppressor *s, float *out, const float *in, size_t n);
size_t afo_suppressor_process_traced(afo_suppressor *s, float *out, const float *in, size_t n, int32_t *pitch, int32_t *silence);
void afo_rnnoise_benchmark_frames(const float *in, float *out, size_t n, uint64_t weight_seed);


/* evaluation order of the long sums: 0 = wavefront-native (the GPU's), 1 = published scalar C (see af_rnnoise.c) */
extern int afo_rnn_eval_order;
/* 0 = the checker's transforms; 1 = packed 480-point mixed-radix transforms for the timed CPU baseline (same results to the
 * last bits; see af_rnnoise.c) */
extern int afo_rnn_fft_mode;
/* RNNoiseProcessor::scale_sample_for_model (rnnoise.rs:89-111) */
float afo_scale_sample_for_model(float sample);

/* RNNoiseProcessor with its fixed rings (rnnoise.rs:11: capacity 8192 + 480) */
#define AFO_RNN_RING_CAPACITY (8192 + 480)
typedef struct {
  float data[AFO_RNN_RING_CAPACITY];
  size_t head, len;
} afo_ring;
typedef struct {
  afo_suppressor core;
  afo_ring input, output;
  int enabled;
} afo_rnnoise_processor;
void afo_processor_init(afo_rnnoise_processor *p, float strength, uint64_t weight_seed);
void afo_processor_set_strength(afo_rnnoise_processor *p, float v);
float afo_processor_get_strength(const afo_rnnoise_processor *p);
size_t afo_processor_push_samples(afo_rnnoise_processor *p, const float *v, size_t n);
void afo_processor_process_frames(afo_rnnoise_processor *p);
size_t afo_processor_available_samples(const afo_rnnoise_processor *p);
size_t afo_processor_pending_input(const afo_rnnoise_processor *p);
size_t afo_processor_read_samples(afo_rnnoise_processor *p, float *out, size_t n);
size_t afo_processor_drain_pending_input(afo_rnnoise_processor *p, float *out, size_t cap);
void afo_processor_set_enabled(afo_rnnoise_processor *p, int on);
void afo_processor_soft_reset(afo_rnnoise_processor *p);
void afo_processor_reset(afo_rnnoise_processor *p);

#ifdef __cplusplus
}
#endif
#endif
