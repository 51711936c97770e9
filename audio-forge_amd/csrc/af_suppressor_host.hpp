// af_suppressor_host.hpp -- host side of the RNNoise suppressor stage: weights, tables, workspace.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "af_suppressor.h"

namespace af {

hipError_t launch_suppressor_analysis(const SuppArgs &a, const SuppTables &tb, hipStream_t stream, hipEvent_t before_pitch);
hipError_t launch_suppressor_synthesis(const SuppArgs &a, const SuppTables &tb, const RnnDeviceWeights &w, hipStream_t stream,
                                       hipEvent_t after_network, hipStream_t finish_stream, hipStream_t network_stream = nullptr,
                                       hipEvent_t after_spectra = nullptr);
hipError_t launch_suppressor_prefilter(const SuppArgs &a, hipStream_t stream);
hipError_t launch_scale_probe(const float *in, float *out, int64_t n, hipStream_t stream);

// int8 network weights in the layout of the public RNNoise model (dense: [in][out]; GRU: [in][3*units],
// gate order z | r | h).  The trained weights of nnnoiseless 0.5.2 are embedded in that crate and are not
// available offline, so engines start from seeded synthetic weights; a real blob (the fifteen arrays
// below, concatenated in declaration order) can be loaded with af_suppressor_load_weights().
struct RnnWeightsI8 {
  int8_t input_dense_w[42 * 24], input_dense_b[24];
  int8_t vad_gru_w[24 * 72], vad_gru_u[24 * 72], vad_gru_b[72];
  int8_t vad_out_w[24 * 1], vad_out_b[1];
  int8_t noise_gru_w[90 * 144], noise_gru_u[48 * 144], noise_gru_b[144];
  int8_t denoise_gru_w[114 * 288], denoise_gru_u[96 * 288], denoise_gru_b[288];
  int8_t denoise_out_w[96 * 22], denoise_out_b[22];
};
static_assert(sizeof(RnnWeightsI8) == 42 * 24 + 24 + 2 * 24 * 72 + 72 + 24 + 1 + 90 * 144 + 48 * 144 + 144 +
                                          114 * 288 + 96 * 288 + 288 + 96 * 22 + 22,
              "weight blob layout");

inline uint64_t splitmix(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
inline void fill_i8(int8_t *dst, size_t n, uint64_t *s, int amp) {
  for (size_t i = 0; i < n; ++i) {
    const int a = (int)(splitmix(s) % (uint64_t)(2 * amp + 1)) - amp;
    const int b = (int)(splitmix(s) % (uint64_t)(2 * amp + 1)) - amp;
    dst[i] = (int8_t)((a + b) / 2);
  }
}
inline void synthetic_weights(RnnWeightsI8 &w, uint64_t seed) {
  uint64_t s = seed;
  fill_i8(w.input_dense_w, sizeof w.input_dense_w, &s, 48);
  fill_i8(w.input_dense_b, sizeof w.input_dense_b, &s, 20);
  fill_i8(w.vad_gru_w, sizeof w.vad_gru_w, &s, 40);
  fill_i8(w.vad_gru_u, sizeof w.vad_gru_u, &s, 40);
  fill_i8(w.vad_gru_b, sizeof w.vad_gru_b, &s, 20);
  fill_i8(w.vad_out_w, sizeof w.vad_out_w, &s, 60);
  fill_i8(w.vad_out_b, sizeof w.vad_out_b, &s, 20);
  fill_i8(w.noise_gru_w, sizeof w.noise_gru_w, &s, 30);
  fill_i8(w.noise_gru_u, sizeof w.noise_gru_u, &s, 30);
  fill_i8(w.noise_gru_b, sizeof w.noise_gru_b, &s, 20);
  fill_i8(w.denoise_gru_w, sizeof w.denoise_gru_w, &s, 24);
  fill_i8(w.denoise_gru_u, sizeof w.denoise_gru_u, &s, 24);
  fill_i8(w.denoise_gru_b, sizeof w.denoise_gru_b, &s, 20);
  fill_i8(w.denoise_out_w, sizeof w.denoise_out_w, &s, 60);
  fill_i8(w.denoise_out_b, sizeof w.denoise_out_b, &s, 40);
}

struct SuppressorHost {
  bool enabled = false;
  bool raw_protocol = false;
  float strength = 1.0f;
  RnnWeightsI8 weights;
  bool weights_dirty = true;
  // device side
  float *d_blob = nullptr;   // all f32 matrices + tables, one allocation
  uint32_t *d_w4 = nullptr;  // the matrices as int8 in the network kernel's operand order (af_suppressor.h)
  size_t blob_floats = 0;
  RnnDeviceWeights dw{};
  SuppTables tables{};
  float *d_state = nullptr;  // [streams][SuppState::kCount]
  float *d_xh = nullptr;
  float *d_ds = nullptr;
  static constexpr int kXhBuffers = 4;    // model-input buffers: the pre-pass runs up to three windows ahead of the synthesis
  static constexpr int kSpecBuffers = 3;  // spectrum / pitch-spectrum / record buffers: the analysis runs up to two windows ahead
  size_t xh_floats = 0;  // floats per model-input buffer
  size_t ws_cells = 0;   // (frame, stream) cells per spectrum / record buffer
  float2 *d_X = nullptr, *d_P = nullptr;
  SuppFrameRec *d_rec = nullptr;
  int ws_frames = 0, ws_streams = 0;

  SuppressorHost() { synthetic_weights(weights, 0x5EEDULL); }

  static void expand_dense(std::vector<float> &dst, const int8_t *w, const int8_t *b, RnnLayerDims d, size_t &off_w,
                           size_t &off_b) {
    off_w = dst.size();
    dst.resize(dst.size() + (size_t)d.k_pad * d.n_pad, 0.0f);
    for (int k = 0; k < d.k_in; ++k)
      for (int n = 0; n < d.n; ++n) dst[off_w + (size_t)k * d.n_pad + n] = (float)w[k * d.n + n];
    off_b = dst.size();
    dst.resize(dst.size() + d.n_pad, 0.0f);
    for (int n = 0; n < d.n; ++n) dst[off_b + n] = (float)b[n];
  }
  static void expand_gru(std::vector<float> &dst, const int8_t *w, const int8_t *u, const int8_t *b, RnnLayerDims d,
                         size_t off_w[3], size_t off_b[3]) {
    const int stride = 3 * d.n;
    for (int g = 0; g < 3; ++g) {
      off_w[g] = dst.size();
      dst.resize(dst.size() + (size_t)d.k_pad * d.n_pad, 0.0f);
      for (int k = 0; k < d.k_in; ++k)
        for (int n = 0; n < d.n; ++n) dst[off_w[g] + (size_t)k * d.n_pad + n] = (float)w[k * stride + g * d.n + n];
      for (int k = 0; k < d.k_rec; ++k)
        for (int n = 0; n < d.n; ++n)
          dst[off_w[g] + (size_t)(d.k_in + k) * d.n_pad + n] = (float)u[k * stride + g * d.n + n];
      off_b[g] = dst.size();
      dst.resize(dst.size() + d.n_pad, 0.0f);
      for (int n = 0; n < d.n; ++n) dst[off_b[g] + n] = (float)b[g * d.n + n];
    }
  }

  hipError_t upload() {
    std::vector<float> blob;
    size_t dw_off, db_off, ow_off, ob_off, vw[3], vb[3], nw[3], nb[3], ew[3], eb[3];
    expand_dense(blob, weights.input_dense_w, weights.input_dense_b, kDimDense, dw_off, db_off);
    expand_gru(blob, weights.vad_gru_w, weights.vad_gru_u, weights.vad_gru_b, kDimVad, vw, vb);
    expand_gru(blob, weights.noise_gru_w, weights.noise_gru_u, weights.noise_gru_b, kDimNoise, nw, nb);
    expand_gru(blob, weights.denoise_gru_w, weights.denoise_gru_u, weights.denoise_gru_b, kDimDenoise, ew, eb);
    expand_dense(blob, weights.denoise_out_w, weights.denoise_out_b, kDimOut, ow_off, ob_off);
    const double pi = 3.14159265358979323846;
    const size_t tansig_off = blob.size();
    for (int i = 0; i <= 200; ++i) blob.push_back((float)std::tanh(0.04 * i));
    while (blob.size() % 4) blob.push_back(0.0f);
    const size_t win_off = blob.size();
    for (int i = 0; i < kRnnFrame; ++i) {
      const double sn = std::sin(.5 * pi * (i + .5) / kRnnFrame);
      blob.push_back((float)std::sin(.5 * pi * sn * sn));
    }
    const size_t dct_off = blob.size();
    for (int i = 0; i < kRnnBands; ++i)
      for (int j = 0; j < kRnnBands; ++j) {
        double v = std::cos((i + .5) * j * pi / kRnnBands);
        if (j == 0) v *= std::sqrt(.5);
        blob.push_back((float)v);
      }
    while (blob.size() % 4) blob.push_back(0.0f);
    const size_t tw_off = blob.size();
    for (int i = 0; i < kRnnWindow; ++i) {
      blob.push_back((float)std::cos(-2.0 * pi * i / kRnnWindow));
      blob.push_back((float)std::sin(-2.0 * pi * i / kRnnWindow));
    }
    while (blob.size() % 4) blob.push_back(0.0f);
    const size_t frac_off = blob.size();
    const int eband[kRnnBands] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40, 48, 60, 78, 100};
    std::vector<int32_t> band_of(484, kRnnBands - 1);
    blob.resize(blob.size() + 404, 0.0f);
    for (int b = 0; b < kRnnBands - 1; ++b) {
      const int size = (eband[b + 1] - eband[b]) << 2;
      for (int j = 0; j < size; ++j) {
        blob[frac_off + (eband[b] << 2) + j] = (float)j / (float)size;
        band_of[(eband[b] << 2) + j] = b;
      }
    }
    const size_t band_off = blob.size();
    blob.resize(blob.size() + 484);
    std::memcpy(&blob[band_off], band_of.data(), 484 * sizeof(int32_t));
    if (blob.size() > blob_floats) {
      if (d_blob) (void)hipFree(d_blob);
      hipError_t err = hipMalloc(&d_blob, blob.size() * sizeof(float));
      if (err != hipSuccess) return err;
      blob_floats = blob.size();
    }
    hipError_t err = hipMemcpy(d_blob, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
    if (err != hipSuccess) return err;
    dw.dense_w = d_blob + dw_off;
    dw.dense_b = d_blob + db_off;
    dw.out_w = d_blob + ow_off;
    dw.out_b = d_blob + ob_off;
    for (int g = 0; g < 3; ++g) {
      dw.vad_w[g] = d_blob + vw[g];
      dw.vad_b[g] = d_blob + vb[g];
      dw.noise_w[g] = d_blob + nw[g];
      dw.noise_b[g] = d_blob + nb[g];
      dw.den_w[g] = d_blob + ew[g];
      dw.den_b[g] = d_blob + eb[g];
    }
    {  // the eleven matrices back as int8 (every entry of the f32 blob is a small integer), same padded layout
      const size_t offs[11] = {dw_off, vw[0], vw[1], vw[2], nw[0], nw[1], nw[2], ew[0], ew[1], ew[2], ow_off};
      const RnnLayerDims dims[11] = {kDimDense, kDimVad, kDimVad, kDimVad, kDimNoise, kDimNoise, kDimNoise,
                                     kDimDenoise, kDimDenoise, kDimDenoise, kDimOut};
      std::vector<int8_t> w8;
      int32_t off8[11];
      for (int i = 0; i < 11; ++i) {
        off8[i] = (int32_t)w8.size();
        const size_t count = (size_t)dims[i].k_pad * dims[i].n_pad;
        for (size_t j = 0; j < count; ++j) w8.push_back((int8_t)blob[offs[i] + j]);
        while (w8.size() % 16) w8.push_back(0);
      }
      // ... re-ordered for the network kernel (af_suppressor.h): k padded to whole groups of 16 with zero weights
      std::vector<uint32_t> w4;
      for (int i = 0; i < 11; ++i) {
        if ((int)w4.size() != w4_matrix_offset(i)) return hipErrorInvalidValue;
        const int n_pad = dims[i].n_pad, tiles = n_pad / 16, groups = (dims[i].k_pad / 4 + 3) / 4;
        for (int g = 0; g < groups; ++g)
          for (int t = 0; t < tiles; ++t)
            for (int lane = 0; lane < 64; ++lane) {
              uint32_t word = 0;
              for (int j = 0; j < 4; ++j) {
                const int k = 16 * g + 4 * j + (lane >> 4), n = 16 * t + (lane & 15);
                const int8_t v = k < dims[i].k_pad ? w8[(size_t)off8[i] + (size_t)k * n_pad + n] : (int8_t)0;
                word |= (uint32_t)(uint8_t)v << (8 * j);
              }
              w4.push_back(word);
            }
      }
      if (!d_w4) {
        hipError_t e4 = hipMalloc(&d_w4, w4.size() * sizeof(uint32_t));
        if (e4 != hipSuccess) return e4;
      }
      hipError_t e4 = hipMemcpy(d_w4, w4.data(), w4.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
      if (e4 != hipSuccess) return e4;
      dw.w4 = d_w4;
    }
    dw.tansig = d_blob + tansig_off;
    tables.half_window = d_blob + win_off;
    tables.dct = d_blob + dct_off;
    tables.twiddle = reinterpret_cast<const float2 *>(d_blob + tw_off);
    tables.frac = d_blob + frac_off;
    tables.band_of_bin = reinterpret_cast<const int32_t *>(d_blob + band_off);
    weights_dirty = false;
    return hipSuccess;
  }

  hipError_t reset_state(int n_streams) {
    if (!d_state) {
      hipError_t err = hipMalloc(&d_state, sizeof(float) * SuppState::kCount * (size_t)n_streams);
      if (err != hipSuccess) return err;
    }
    std::vector<float> row(SuppState::kCount, 0.0f);
    row[SuppState::kSmoothedStrength] = 1.0f;  // rnnoise.rs:59
    std::vector<float> all((size_t)SuppState::kCount * n_streams);
    for (int s = 0; s < n_streams; ++s) std::memcpy(&all[(size_t)s * SuppState::kCount], row.data(), sizeof(float) * row.size());
    return hipMemcpy(d_state, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice);
  }

  hipError_t ensure_workspace(int n_streams, int frames) {
    if (frames <= ws_frames && n_streams == ws_streams) return hipSuccess;
    release_workspace();
    hipError_t err;
    const size_t cells = (size_t)frames * n_streams;
    // The windows of a call run as a pipeline (pre-pass | analysis | synthesis), so the buffers that cross a
    // stage boundary exist several times: the model input lives from the pre-pass to the synthesis (3 windows
    // in flight), spectra and frame records from the analysis to the synthesis (2), pitch spectra / resynthesised
    // frames from the network half of the synthesis to its resynthesis half (2).
    xh_floats = (size_t)n_streams * (kPitchBuf + (size_t)frames * kRnnFrame);
    ws_cells = cells;
    if ((err = hipMalloc(&d_xh, sizeof(float) * kXhBuffers * xh_floats)) != hipSuccess) return err;
    if ((err = hipMalloc(&d_X, sizeof(float2) * kSpecBuffers * cells * kRnnFreq)) != hipSuccess) return err;
    if ((err = hipMalloc(&d_P, sizeof(float2) * kSpecBuffers * cells * kRnnFreq)) != hipSuccess) return err;
    if ((err = hipMalloc(&d_rec, sizeof(SuppFrameRec) * kSpecBuffers * cells)) != hipSuccess) return err;
    if ((err = hipMalloc(&d_ds, sizeof(float) * cells * (kPitchBuf / 2))) != hipSuccess) return err;
    ws_frames = frames;
    ws_streams = n_streams;
    return hipSuccess;
  }
  void release_workspace() {
    (void)hipFree(d_xh);
    (void)hipFree(d_ds);
    d_ds = nullptr;
    (void)hipFree(d_X);
    (void)hipFree(d_P);
    (void)hipFree(d_rec);
    d_xh = nullptr;
    d_X = d_P = nullptr;
    d_rec = nullptr;
    ws_frames = ws_streams = 0;
  }
  void release_all() {
    release_workspace();
    (void)hipFree(d_blob);
    (void)hipFree(d_w4);
    d_w4 = nullptr;
    (void)hipFree(d_state);
    d_blob = d_state = nullptr;
    blob_floats = 0;
  }
};

}  // namespace af
