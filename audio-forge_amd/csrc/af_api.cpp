// af_api.cpp -- the C ABI of include/audioforge_mi.h on top of the host mirror and the kernels.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/audioforge_mi.h"
#include "af_device.h"
#include "af_host.hpp"
#include "af_resampler_host.hpp"
#include "af_stages.h"
#include "af_suppressor_host.hpp"

namespace af {
size_t lane_kernel_dynamic_lds(int lookahead_samples);
hipError_t launch_chain_lane(const LaunchArgs &args, int lookahead_samples, hipStream_t stream);
size_t ring_kernel_dynamic_lds(int n_sections, int lookahead_samples, bool crossfade);
hipError_t launch_chain_ring(const LaunchArgs &args, int n_sections, int lookahead_samples, bool crossfade, int variant,
                             bool auto_makeup, hipStream_t stream);
hipError_t launch_chain_ring_lds(const LaunchArgs &args, size_t dyn, int variant, bool auto_makeup, hipStream_t stream);
hipError_t launch_chain_publish_ready(int64_t *ready, int64_t samples, hipStream_t stream);
hipError_t launch_eq_stream_part(const ChainParams *d_params, double *st64, const float *in, float *audio, BlockStats *stats,
                                 double *block_power, int sec0, int count, bool head, int64_t n_samples, int64_t stream_stride,
                                 int32_t n_streams, hipStream_t stream);
hipError_t launch_chain_quad(const LaunchArgs &args, int n_sections, int lookahead_samples, bool crossfade, int waves,
                             hipStream_t stream);
size_t quad_kernel_dynamic_lds(int n_sections, int lookahead_samples, bool crossfade);
hipError_t launch_merge_side_stats(BlockStats *rows, const BlockStats *input_rows, const BlockStats *deesser_rows,
                                   int64_t n, hipStream_t stream);
hipError_t launch_resample(const double *in, double *out, const ResamplePos *pos, const double *table, int64_t n_in,
                           int64_t n_out, int64_t in_stride, int64_t out_stride, int32_t n_streams, int32_t sinc_len,
                           double ratio, int variant, hipStream_t stream);
int resample_segment_outputs(double ratio, int sinc_len);
hipError_t launch_kweight_energy(const float *audio, double *partial, int32_t *non_finite, const double b[5],
                                 const double a5[5], int64_t n_samples, int64_t stride, int64_t n100, int32_t n_streams,
                                 int32_t s100, hipStream_t stream);
hipError_t launch_deesser(const ChainParams *d_params, double *st64, float *st32, const float *in, float *out,
                          BlockStats *rows, int64_t n_samples, int64_t stream_stride, int32_t n_streams,
                          int32_t layout, bool front_end, bool write_out_power, hipStream_t stream);
hipError_t launch_eq_systolic(const ChainParams *d_params, const int32_t *d_group_preset, double *st64, const float *in, float *audio,
                              float *ring, float *ring_in, int32_t ring_rows, int64_t n0, BlockStats *stats, bool crossfade,
                              int64_t n_samples, int64_t stream_stride, int32_t n_streams, hipStream_t stream,
                              double *block_power = nullptr, int n_sections = -1);
bool comp_roles_serves(const ChainParams &p);
bool lim_roles_serves(const ChainParams &p);
hipError_t launch_chain_comp_roles(const LaunchArgs &args, bool sidechain, bool adaptive, hipStream_t stream);
hipError_t launch_chain_lim_roles(const LaunchArgs &args, int max_lookahead, hipStream_t stream);
constexpr size_t kMaxLdsBytes = 160 * 1024;
}  // namespace af

// AUTO routes the configurations the stage pipeline serves to it up to this many streams.  Measured (dynamics chain, 10 s,
// one MI355X): 256 streams 41 ms against the token ring's 202, 1024 streams 64, 2048 streams 91, 3072 streams 136, 4096
// streams 202 (the hand-over rings cost 0.3 KB per sample step per stream: the HBM roof); beyond that the token ring wins
// (a launch of it lasts ~202 ms up to 16 384 streams).  Behind the suppressor, whose kernels use the same HBM, up to 2048.
constexpr int kStagedAutoMaxStreams = 3072, kStagedAutoMaxStreamsBehindSuppressor = 2048;
constexpr int kEqParamSlots = 16;  // parameter blocks the EQ / de-esser stages read: one per window in flight (af_engine::d_params_eq)

static_assert(sizeof(af_block_stats) == sizeof(af::BlockStats), "stats row layout");
static_assert(sizeof(af_block_stats) == 72, "stats row size");

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define AF_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t err__ = (expr);                                                                    \
    if (err__ != hipSuccess)                                                                      \
      return fail(AF_ERR_BACKEND, "%s failed: %s", #expr, hipGetErrorString(err__));              \
  } while (0)

}  // namespace

// one more chain preset of an engine (preset 0 is af_engine::proto / host_params)
struct af_preset {
  af::ChainProto proto;
  af::ChainParams host_params{};
  explicit af_preset(double fs) : proto(fs) {}
};

struct af_engine {
  af::ChainProto proto;
  // Presets.  An engine is N streams in groups of 64 (one chain workgroup each); every group runs one preset.  Preset 0 is
  // `proto`; af_engine_set_preset_count adds fresh ones, the setters address the selected one, af_engine_assign_presets
  // maps groups to presets.  The device holds the parameter blocks as an array and a group -> preset table.
  std::vector<af_preset> extra_presets;
  int current_preset = 0;
  std::vector<int32_t> group_preset;         // [ceil(n_streams / 64)], empty = every group runs preset 0
  int32_t *d_group_preset = nullptr;
  af::ChainParams *d_params_multi = nullptr;  // [1 + extra_presets.size()]
  af::ChainParams *d_params_eq = nullptr;     // [1 + extra_presets.size()]: what the systolic EQ kernel reads (af_eq_systolic.hip)
  std::vector<af::ChainParams> uploaded_eq;
  int eq_params_presets = 0;
  uint64_t eq_slot_cursor = 0;               // stage pipeline with the de-esser: the next window's parameter slot
  std::vector<af::ChainParams> uploaded_multi;
  int n_streams;
  int device;
  bool started = false;
  bool params_dirty = true;
  int kernel = AF_KERNEL_AUTO;
  int ring_variant = 0;
  bool timing = false;
  int64_t samples_processed = 0;
  int64_t last_blocks = 0;
  double last_kernel_ms = 0.0;
  int last_launches = 0;
  int last_kernel_used = 0;  // AF_KERNEL_* of the most recent chain launch

  af::ChainParams host_params{};
  af::ChainParams uploaded{};   // what d_params currently holds
  bool uploaded_valid = false;
  af::ChainParams *d_params = nullptr;
  double *d_st64 = nullptr;
  float *d_st32 = nullptr;
  int n_f64 = 0, n_f32 = 0;
  af::BlockStats *d_stats = nullptr;
  af::BlockStats *d_stats_pre = nullptr;   // rows of the pre-pass launch (auto-makeup)
  int64_t stats_pre_capacity = 0;
  af::ChainParams *d_params_pre = nullptr;
  af::ChainParams uploaded_pre{};          // what d_params_pre currently holds
  bool uploaded_pre_valid = false;
  double *d_block_power = nullptr;         // [blocks][streams] of the current call: compressor-input block power written by the systolic EQ
  int64_t block_power_capacity = 0;        // doubles
  // Parameter uploads go through engine-owned pinned staging slots (stage_upload): the host never waits for a stream, and
  // a slot is only reused once the copy that read it has run.
  struct ParamStager {
    static constexpr int kSlots = 32;      // (the stage pipeline with the de-esser uploads one block per window: the host may run this many windows ahead)
    af::ChainParams *pinned = nullptr;     // [kSlots][blocks_per_slot]
    size_t blocks_per_slot = 0;
    hipEvent_t done[kSlots] = {};
    bool used[kSlots] = {};
    int next = 0;
  } stager;
  // Device buffers that had to grow while earlier work may still read them: kept until that work has ended (an event on the
  // stream the call was made on), freed by a later call, a reset or the destructor -- growing never synchronises the device.
  struct Retired { void *p; hipEvent_t ev; };
  std::vector<Retired> retired;
  hipStream_t syn_stream = nullptr;        // CU partition: pitch spectra + network + resynthesis (else the caller's stream)
  hipStream_t fin_stream = nullptr;        // resynthesis + overlap-add of window w beside pitch spectra + network of w+1
  hipStream_t rnn_stream = nullptr;        // the network of window w beside the pitch spectra of w+1 (AF_RNN_STREAM=0: on syn_stream)
  hipStream_t lim_stream = nullptr;        // AF_ROLES=2: the limiter half of the chain (af_roles.hip) on CUs of its own
  hipStream_t eq_stream = nullptr;         // the window's systolic EQ (af_eq_systolic.hip), behind its overlap-add, beside the next window's synthesis
  int partition_chain_cus = 0;             // CUs reserved for the chain stream (0 = the streams are not masked)
  af::BlockStats *d_stats_de = nullptr;    // rows of the de-esser pass
  af::ChainParams *d_params_de = nullptr;  // the de-esser pass reads the unedited parameter block
  af::ChainParams uploaded_de{};
  bool uploaded_de_valid = false;
  double *d_vad = nullptr;                 // [blocks][streams] speech posteriors for the next call
  int64_t vad_capacity = 0, vad_blocks = 0;
  double vad_reliability = 0.0, noise_floor_db = 0.0, live_noise_reliability = 0.0;
  bool has_evidence = false;
  int32_t *d_status = nullptr;
  bool eq_params_on_es = false;              // the EQ parameter block's last upload ran on the EQ stream (two-part EQ: who must wait for it)
  int64_t *d_ready = nullptr;                // samples of the running call the suppressor's side has finished (LaunchArgs::ready)
  int64_t stats_capacity = 0;  // rows
  float *d_io = nullptr;       // staging for the host entry point
  int64_t io_capacity = 0;     // floats
  hipStream_t last_stream = nullptr;
  af::SuppressorHost supp;
  bool borrowed_streams = false;                         // AF_SERIAL_STREAMS: the side streams alias the caller's
  hipStream_t aux_stream = nullptr;                      // chain launches while the suppressor fills the chip
  hipStream_t pre_stream = nullptr;                      // the suppressor's sample-serial pre-pass, two windows ahead
  hipStream_t ana_stream = nullptr;                      // spectra + pitch, one window ahead
  std::vector<hipEvent_t> sync_events;
  size_t ev_cursor = 0;                                  // next free entry of sync_events within the current call
  std::vector<std::pair<hipEvent_t, hipEvent_t>> chain_ms_events;  // timing brackets of the chain launches of the last call
  // rnnoise.rs:114-164: the samples of a call that do not fill a 480-sample frame wait here for the next call
  float *d_pending = nullptr;   // [streams][480]
  int pending = 0;              // samples per stream waiting in d_pending (all streams advance in lock step)
  float *d_asm = nullptr;       // [streams][asm_stride]: pending samples + this call's, when the two have to be joined
  int64_t asm_capacity = 0;     // floats
  int64_t last_output_samples = 0;  // samples per stream the last process call produced
  int32_t *d_trace = nullptr;   // [frames][streams][2]: (silence, pitch index) of every frame of the last call
  int64_t trace_capacity = 0, trace_frames = 0;
  bool trace = false;
  // the stage-pipeline form of the chain (af_stages.hip): rings, one stream and a ring of events per stage
  struct StagePipe {
    bool decided = false, active = false;  // chosen at the first call after a reset, then kept (the rings ARE the histories)
    af::StageRings rings{};
    std::vector<void *> allocs;
    int64_t tw_max = 0;                    // longest window the rings were sized for
    hipStream_t stream = nullptr;          // where the launch steps go when the suppressor's pipeline feeds the chain
    int64_t windows = 0;                   // windows launched since the rings were last cleared
    static constexpr int kMkSets = 4;
    double *d_mk = nullptr;
    int64_t mk_rows = 0;                   // rows (blocks x streams) per set
    static constexpr int kBpSets = 16;
    double *d_bp = nullptr;                // auto-makeup: block powers, written seven launch steps before they are read
    int64_t call_stride = 0;               // stream stride of the call being scheduled
    const af::ChainParams *d_chain = nullptr;  // the parameter block(s) the stages read (an array with several presets)
    bool with_deesser = false;             // the rings include the de-esser stages'
    int32_t w_min = 1;                     // smallest lookahead + 1 over the presets
    uint32_t strip = 0;                    // chain flags another kernel has taken over (the suppressor's pre-pass)
  } pipe;
  // measured 12..80 (AF_SUPP_WINDOW_FRAMES).  Round 1 (one chain launch per window): 20 -> 248 ms per bench step, 24 -> 252, 30 ->
  // 254, 16 -> 259, 50 -> 260.  End of round 3 (one chain launch per call: a window costs the chain nothing any more):
  // 12 -> 165.2, 14 -> 166.1, 16 -> 163.2, 18 -> 164.5, 20 -> 165.0, 24 -> 166.7, 32 -> 169.2
  int supp_window_frames = 16;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_mid = nullptr;  // start | suppressor done | chain done

  af_engine(double fs, int n, int dev) : proto(fs), n_streams(n), device(dev) {}
};

namespace {

int require_config(af_engine *e) {
  if (!e) return fail(AF_ERR_INVALID_ARGUMENT, "engine is null");
  if (e->started)
    return fail(AF_ERR_STATE, "setter called after streaming started; call af_engine_reset first");
  e->params_dirty = true;
  return AF_OK;
}

// Copy `count` parameter blocks to the device behind everything already queued on `stream`, from a pinned engine-owned slot.
int stage_upload(af_engine *e, af::ChainParams *dst, const af::ChainParams *src, size_t count, hipStream_t stream) {
  auto &st = e->stager;
  if (count > st.blocks_per_slot) {
    for (int k = 0; k < af_engine::ParamStager::kSlots; ++k)
      if (st.used[k]) { AF_HIP(hipEventSynchronize(st.done[k])); st.used[k] = false; }
    if (st.pinned) AF_HIP(hipHostFree(st.pinned));
    st.pinned = nullptr;
    AF_HIP(hipHostMalloc(reinterpret_cast<void **>(&st.pinned), sizeof(af::ChainParams) * count * af_engine::ParamStager::kSlots, hipHostMallocDefault));
    st.blocks_per_slot = count;
  }
  const int slot = st.next;
  st.next = (st.next + 1) % af_engine::ParamStager::kSlots;
  if (!st.done[slot]) AF_HIP(hipEventCreateWithFlags(&st.done[slot], hipEventDisableTiming));
  if (st.used[slot]) AF_HIP(hipEventSynchronize(st.done[slot]));  // kSlots uploads ago: normally long done
  af::ChainParams *host = st.pinned + (size_t)slot * st.blocks_per_slot;
  std::memcpy(host, src, sizeof(af::ChainParams) * count);
  AF_HIP(hipMemcpyAsync(dst, host, sizeof(af::ChainParams) * count, hipMemcpyHostToDevice, stream));
  AF_HIP(hipEventRecord(st.done[slot], stream));
  st.used[slot] = true;
  return AF_OK;
}

// free the retired buffers whose last reader has ended (`all`: wait for them)
int collect_retired(af_engine *e, bool all) {
  size_t kept = 0;
  for (auto &r : e->retired) {
    hipError_t q = all ? hipEventSynchronize(r.ev) : hipEventQuery(r.ev);
    if (q == hipSuccess) {
      (void)hipFree(r.p);
      (void)hipEventDestroy(r.ev);
    } else {
      if (q != hipErrorNotReady) (void)hipGetLastError();
      e->retired[kept++] = r;
    }
  }
  e->retired.resize(kept);
  return AF_OK;
}

// Make `*p` hold at least `need` bytes.  Growth is geometric; the old buffer (its contents are per-call scratch, never
// carried over) is retired behind an event on `stream` instead of being freed under the feet of queued kernels.
int grow_device(af_engine *e, void **p, int64_t *capacity_bytes, int64_t need, hipStream_t stream) {
  if (need <= *capacity_bytes) return AF_OK;
  const int64_t cap = std::max<int64_t>(need, *capacity_bytes + *capacity_bytes / 2);
  void *fresh = nullptr;
  AF_HIP(hipMalloc(&fresh, (size_t)cap));
  if (*p) {
    hipEvent_t ev;
    AF_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    AF_HIP(hipEventRecord(ev, stream));
    e->retired.push_back({*p, ev});
  }
  *p = fresh;
  *capacity_bytes = cap;
  return AF_OK;
}

// the prototype the setters address
af::ChainProto &cur(af_engine *e) { return e->current_preset == 0 ? e->proto : e->extra_presets[e->current_preset - 1].proto; }
const af::ChainProto &cur(const af_engine *e) { return e->current_preset == 0 ? e->proto : e->extra_presets[e->current_preset - 1].proto; }

int check_band(int32_t band) {
  if (band < 0 || band >= af::kNumBands) return fail(AF_ERR_INVALID_ARGUMENT, "band index %d out of range", band);
  return AF_OK;
}

af::EqBandConfig to_cfg(const af_eq_band_config &c) {
  return af::EqBandConfig{c.filter_type, c.frequency_hz, c.gain_db, c.q, c.slope_db_per_octave, c.enabled != 0};
}

// Flatten the prototype into ChainParams (uniform) ...
void export_params(af_engine *e, const af::ChainProto &p, af::ChainParams &o) {
  std::memset(&o, 0, sizeof o);
  uint32_t f = 0;
  if (p.deesser_enabled) f |= af::kFlagDeesser;
  if (p.eq_enabled) f |= af::kFlagEq;
  if (p.compressor_enabled) f |= af::kFlagCompressor;
  if (p.limiter_enabled) f |= af::kFlagLimiter;
  if (p.eq_before_deesser) f |= af::kFlagEqBeforeDeesser;
  if (p.input_scrub) f |= af::kFlagInputScrub;
  if (p.input_clamp) f |= af::kFlagInputClamp;
  if (p.dc_block) f |= af::kFlagDcBlock;
  if (p.pre_highpass) f |= af::kFlagPreHighpass;
  o.flags = f;
  o.control_block = p.control_block;
  o.pre_hp = af::rbj_coefficients(af::BiquadType::HighPass, 80.0, 0.0, 0.707, p.sample_rate);
  int n = 0;
  for (int b = 0; b < af::kNumBands; ++b)
    for (int s = 0; s < p.eq.bands[b].processing_sections; ++s) o.eq[n++] = p.eq.bands[b].sections[s].section();
  o.n_eq_sections = n;
  o.comp = p.compressor.params(p.control_block);
  o.comp.vad_reliability = e->vad_reliability;
  o.comp.noise_floor_db = e->noise_floor_db;
  o.comp.live_noise_reliability = e->live_noise_reliability;
  o.comp.has_evidence = e->has_evidence ? 1 : 0;
  o.lim = p.limiter.params();
  o.deesser = p.deesser.params();
  // block_processor.rs:150-151: the TP ceiling follows the limiter ceiling on every block
  af::TruePeakProto tp = p.tp_limiter;
  tp.set_ceiling_linear(std::pow(10.0f, (float)p.limiter.ceiling_db / 20.0f));
  o.tp.ceiling_linear = tp.ceiling_linear;
  o.tp.release_coeff = tp.release_coeff;
}
void export_params(af_engine *e) {
  export_params(e, e->proto, e->host_params);
  for (af_preset &ps : e->extra_presets) export_params(e, ps.proto, ps.host_params);
}
const af::ChainProto &preset_proto(const af_engine *e, int k) { return k == 0 ? e->proto : e->extra_presets[k - 1].proto; }
af::ChainParams &preset_params(af_engine *e, int k) { return k == 0 ? e->host_params : e->extra_presets[k - 1].host_params; }
int preset_of_stream(const af_engine *e, int64_t s) { return e->group_preset.empty() ? 0 : e->group_preset[s / 64]; }

// ... and the initial per-stream state planes.
int upload_initial_state(af_engine *e) {
  const int64_t B = e->n_streams;
  const int n_presets = 1 + (int)e->extra_presets.size();
  int n64 = 0, n32 = 0;
  for (int k = 0; k < n_presets; ++k) {
    const af::ChainProto &p = preset_proto(e, k);
    const af::ChainParams &hp = preset_params(e, k);
    const int meter_slots = (p.compressor_enabled && p.compressor.auto_makeup_enabled) ? hp.comp.meter_slots : 0;
    n64 = std::max(n64, af::f64_field_count(hp.n_eq_sections, meter_slots));
    n32 = std::max(n32, af::f32_field_count(p.limiter.lookahead_samples));
  }
  if (e->d_st64 && (n64 != e->n_f64 || n32 != e->n_f32)) {
    AF_HIP(hipFree(e->d_st64));
    AF_HIP(hipFree(e->d_st32));
    e->d_st64 = nullptr;
    e->d_st32 = nullptr;
  }
  if (!e->d_st64) {
    AF_HIP(hipMalloc(&e->d_st64, sizeof(double) * n64 * B));
    AF_HIP(hipMalloc(&e->d_st32, sizeof(float) * n32 * B));
    e->n_f64 = n64;
    e->n_f32 = n32;
  }
  // every stream starts from its preset's prototype
  std::vector<std::vector<double>> v64s(n_presets, std::vector<double>(n64, 0.0));
  std::vector<std::vector<float>> v32s(n_presets, std::vector<float>(n32, 0.0f));
  for (int k = 0; k < n_presets; ++k) {
    std::vector<double> &v64 = v64s[k];
    std::vector<float> &v32 = v32s[k];
    const af::CompressorProto &c = preset_proto(e, k).compressor;
    v64[af::kCompPeakEnvDb] = -120.0;
    v64[af::kCompGr] = c.current_gain_reduction_db;
    v64[af::kCompFastEnv] = c.fast_release_env_db;
    v64[af::kCompSlowEnv] = c.slow_release_env_db;
    v64[af::kCompCurReleaseMs] = c.current_release_ms;
    v64[af::kCompTargetReleaseMs] = c.target_release_ms;
    // compressor.rs:760-761: release_coeff is tc(current_release_ms) from the first sample on
    v64[af::kCompReleaseCoeff] = af::time_constant_to_coeff(c.current_release_ms, c.sample_rate);
    v64[af::kCompSmoothedMakeup] = c.smoothed_makeup_gain;
    v64[af::kCompCurrentLufs] = -100.0;
    v64[af::kLimGain] = 1.0;
    for (int i = 0; i < 3; ++i) {  // the dynamic EQ's coefficients are per-stream state (deesser.rs:536-538)
      const af::BiquadCoef &bc = preset_params(e, k).deesser.bands[i].dynamic_eq.active;
      double *d = &v64[af::kDeBand0 + i * af::kDeBandStride + 6];
      d[0] = bc.b0; d[1] = bc.b1; d[2] = bc.b2; d[3] = bc.a1; d[4] = bc.a2;
    }
    v32[af::kTpGain] = 1.0f;
  }
  std::vector<double> plane64((size_t)n64 * B);
  std::vector<float> plane32((size_t)n32 * B);
  for (int64_t st = 0; st < B; ++st) {
    const int k = preset_of_stream(e, st);
    for (int f = 0; f < n64; ++f) plane64[(size_t)f * B + st] = v64s[k][f];
    for (int f = 0; f < n32; ++f) plane32[(size_t)f * B + st] = v32s[k][f];
  }
  AF_HIP(hipMemcpy(e->d_st64, plane64.data(), plane64.size() * sizeof(double), hipMemcpyHostToDevice));
  AF_HIP(hipMemcpy(e->d_st32, plane32.data(), plane32.size() * sizeof(float), hipMemcpyHostToDevice));
  return AF_OK;
}

int ensure_started(af_engine *e) {
  AF_HIP(hipSetDevice(e->device));
  if (!e->started) {
    export_params(e);
    // the limiter's delay ring and suffix maxima live in LDS: kernel 3 (16 streams per workgroup) holds ~1000 samples
    // of lookahead (2 ms at 384 kHz = 768), kernels 1 and 2 (64 streams) ~127 / ~180
    if (e->proto.limiter_enabled &&
        af::quad_kernel_dynamic_lds(e->host_params.n_eq_sections, e->proto.limiter.lookahead_samples, true) > af::kMaxLdsBytes)
      return fail(AF_ERR_UNSUPPORTED, "limiter lookahead of %d samples exceeds what the LDS-resident ring holds",
                  e->proto.limiter.lookahead_samples);
    if (e->proto.compressor_enabled && e->proto.compressor.auto_makeup_enabled) {
      if (e->host_params.comp.meter_slots <= 0)
        return fail(AF_ERR_UNSUPPORTED, "auto-makeup needs a control block that divides the 400 ms loudness window "
                                        "(e.g. 480 or 960 samples at 48 kHz) and a sample rate the meter supports");
      if (e->host_params.control_block < 64)
        return fail(AF_ERR_UNSUPPORTED, "auto-makeup needs control blocks of at least 64 samples");
    }
    if (!e->extra_presets.empty()) {
      // several presets in one engine: the plain one-launch form of the token-ring kernel serves them (the workgroup of a
      // 64-stream group reads its own parameter block); what shapes the launch itself must agree across presets
      for (const af_preset &ps : e->extra_presets) {
        if (ps.host_params.control_block != e->host_params.control_block)
          return fail(AF_ERR_INVALID_ARGUMENT, "every preset of an engine must use the same control block");
        if (ps.proto.sample_rate != e->proto.sample_rate) return fail(AF_ERR_INVALID_ARGUMENT, "presets must share the sample rate");
      }
      if (e->kernel != AF_KERNEL_AUTO && e->kernel != AF_KERNEL_PHASED && e->kernel != AF_KERNEL_STAGED)
        return fail(AF_ERR_UNSUPPORTED, "multi-preset engines run the token-ring kernel or the stage pipeline");
      const size_t n_groups = (size_t)(e->n_streams + 63) / 64;
      if (e->group_preset.size() != n_groups) e->group_preset.assign(n_groups, 0);
      if (e->d_group_preset) AF_HIP(hipFree(e->d_group_preset));
      if (e->d_params_multi) AF_HIP(hipFree(e->d_params_multi));
      e->d_group_preset = nullptr;
      e->d_params_multi = nullptr;
      AF_HIP(hipMalloc(&e->d_group_preset, sizeof(int32_t) * n_groups));
      AF_HIP(hipMemcpy(e->d_group_preset, e->group_preset.data(), sizeof(int32_t) * n_groups, hipMemcpyHostToDevice));
      AF_HIP(hipMalloc(&e->d_params_multi, sizeof(af::ChainParams) * (1 + e->extra_presets.size())));
      e->uploaded_multi.clear();
    }
    if (!e->d_params) AF_HIP(hipMalloc(&e->d_params, sizeof(af::ChainParams)));
    if (!e->d_params_pre) AF_HIP(hipMalloc(&e->d_params_pre, sizeof(af::ChainParams)));
    if (!e->d_params_de) AF_HIP(hipMalloc(&e->d_params_de, sizeof(af::ChainParams)));
    if (!e->d_status) {
      AF_HIP(hipMalloc(&e->d_status, sizeof(int32_t)));
      AF_HIP(hipMemset(e->d_status, 0, sizeof(int32_t)));
    }
    int rc = upload_initial_state(e);
    if (rc) return rc;
    if (e->supp.enabled) {
      if (e->proto.sample_rate != 48000.0)
        return fail(AF_ERR_INVALID_ARGUMENT, "the RNNoise suppressor runs at 48 kHz only (rnnoise.rs:3,46)");
      AF_HIP(e->supp.reset_state(e->n_streams));
    }
    e->params_dirty = true;
    e->started = true;
    e->samples_processed = 0;
    e->pending = 0;
  }
  return AF_OK;
}

int check_device_status(af_engine *e) {
  if (!e->d_status) return AF_OK;
  int32_t st = 0;
  AF_HIP(hipMemcpy(&st, e->d_status, sizeof st, hipMemcpyDeviceToHost));
  if (st != 0)
    return fail(AF_ERR_BACKEND, "a chain kernel abandoned a stage token (device status %d); results are invalid.  (A one-launch call "
                                "waits for kernels on other streams: if something serialises dispatches, set AF_CHAIN_PERSISTENT=0.)", st);
  return AF_OK;
}

// AF_ROLES (same-box A/B): 0 = AUTO never takes the role kernels; 1 = compressor and limiter both as role kernels;
// 2 = the compressor stays on the token-ring kernel (which then ends at the compressor's output) and the limiter half runs as
// the role kernel -- behind the suppressor on a stream and CUs of its own, one window behind the compressor.
int roles_mode() {
  static const int mode = [] {
    const char *env = std::getenv("AF_ROLES");
    return env ? std::atoi(env) : 0;
  }();
  return mode;
}

// AF_CHAIN_PERSISTENT=0 / 1; default on, except under tools that serialise dispatches: a launch that waits for kernels on other
// streams needs them to run beside it (counter collection of rocprofv3 `--pmc`, the runtime's blocking-launch debug switches).
bool one_launch_calls_enabled() {
  static const bool on = [] {
    const char *env = std::getenv("AF_CHAIN_PERSISTENT");
    if (env) return std::atoi(env) != 0;
    for (const char *name : {"ROCPROF_COUNTER_COLLECTION", "AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING", "CUDA_LAUNCH_BLOCKING"}) {
      const char *v = std::getenv(name);
      if (v && std::atoi(v) != 0) return false;
    }
    return true;
  }();
  return on;
}

// after a launch of n samples: advance the (stream-uniform) crossfade counters
void advance_crossfades(af_engine *e, int64_t n) {
 for (int preset = 0; preset <= (int)e->extra_presets.size(); ++preset) {
  af::ChainParams &o = preset_params(e, preset);
  std::vector<af::SectionParams *> sections;
  for (int k = 0; k < o.n_eq_sections; ++k) sections.push_back(&o.eq[k]);
  if (o.flags & af::kFlagDeesser)
    for (auto &b : o.deesser.bands) {
      sections.push_back(&b.detector_hp);
      sections.push_back(&b.detector_lp);
      sections.push_back(&b.dynamic_eq);
    }
  for (af::SectionParams *spp : sections) {
    af::SectionParams &sp = *spp;
    if (sp.xf_remaining > 0) {
      if (n >= sp.xf_remaining) {  // promote_pending_coefficients, biquad.rs:276-286
        sp.active = sp.pending;
        sp.xf_remaining = 0;
        sp.xf_total = 0;
      } else {
        sp.xf_remaining -= (int)n;
      }
      e->params_dirty = true;
    }
  }
 }
}

// Several presets in one engine: one launch of the token-ring kernel, the workgroup of every 64-stream group reading its
// own parameter block.  `strip` = flags the caller's pipeline has already taken care of (the suppressor's front end).
int launch_chain_multi(af_engine *e, uint32_t strip, uint32_t add, const float *in, float *out, int64_t n_samples,
                       int64_t stream_stride, int32_t layout, int64_t samples_before, af::BlockStats *stats, hipStream_t stream,
                       bool stats_cleared) {
  const int n_presets = 1 + (int)e->extra_presets.size();
  for (int k = 0; k < n_presets; ++k) {
    // (the stage pipeline runs both for several presets, when the presets agree on which stages there are: stage_pipe_serves)
    const af::ChainParams &hp = preset_params(e, k);
    if ((hp.flags & af::kFlagDeesser) || ((hp.flags & af::kFlagCompressor) && hp.comp.auto_makeup_enabled))
      return fail(AF_ERR_UNSUPPORTED, "several presets with the de-esser or auto-makeup run on the stage pipeline only, and there every "
                                      "preset must enable the same stages (de-esser ahead of the EQ, compressor, auto-makeup, limiter)");
  }
  std::vector<af::ChainParams> runs((size_t)n_presets);
  size_t dyn = 0;  // every workgroup lays out its own preset: the launch needs the largest of the layouts
  for (int k = 0; k < n_presets; ++k) {
    runs[k] = preset_params(e, k);
    runs[k].flags = (runs[k].flags & ~strip) | add;
    bool xf = false;
    for (int j = 0; j < runs[k].n_eq_sections; ++j) xf |= runs[k].eq[j].xf_remaining > 0;
    dyn = std::max(dyn, af::ring_kernel_dynamic_lds(runs[k].n_eq_sections, runs[k].lim.lookahead_samples, xf));
  }
  if (dyn > af::kMaxLdsBytes)
    return fail(AF_ERR_UNSUPPORTED, "the token-ring kernel needs more LDS than a CU has for one of the presets");
  e->last_kernel_used = AF_KERNEL_PHASED;
  const int cb = runs[0].control_block;
  const int64_t rows = ((n_samples + cb - 1) / cb) * e->n_streams;
  if (e->uploaded_multi.size() != runs.size() ||
      std::memcmp(e->uploaded_multi.data(), runs.data(), sizeof(af::ChainParams) * runs.size()) != 0) {
    e->uploaded_multi = runs;  // (what the device holds; the copy itself reads a pinned slot)
    if (int rc = stage_upload(e, e->d_params_multi, runs.data(), runs.size(), stream)) return rc;
  }
  af::LaunchArgs a{};
  a.st64 = e->d_st64;
  a.st32 = e->d_st32;
  a.in = in;
  a.out = out;
  a.stats = stats;
  a.status = e->d_status;
  a.params = e->d_params_multi;
  a.group_preset = e->d_group_preset;
  a.n_samples = n_samples;
  a.stream_stride = stream_stride;
  a.samples_before = samples_before;
  a.n_streams = e->n_streams;
  a.layout = layout;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  if (e->timing) {
    AF_HIP(hipEventCreate(&t0));
    AF_HIP(hipEventCreate(&t1));
    AF_HIP(hipEventRecord(t0, stream));
  }
  if (!stats_cleared) AF_HIP(hipMemsetAsync(stats, 0, sizeof(af::BlockStats) * rows, stream));  // fields are written by their tokens
  AF_HIP(af::launch_chain_ring_lds(a, dyn, e->ring_variant, false, stream));
  e->last_launches += 1;
  if (e->timing) {
    AF_HIP(hipEventRecord(t1, stream));
    e->chain_ms_events.push_back({t0, t1});
  }
  advance_crossfades(e, n_samples);
  return AF_OK;
}

int engine_event(af_engine *e, hipEvent_t *out_ev);

// One pass of the chain over a segment of samples for every stream: one launch, or the pre-pass + main
// pair when the compressor's auto-makeup needs whole-block input power first (compressor.rs:710).
// `params_stream` is where parameter uploads are ordered; `stream` is where the kernels run.
int launch_chain_segment(af_engine *e, const af::ChainParams &run_in, bool run_modified, const float *in, float *out,
                         int64_t n_samples, int64_t stream_stride, int32_t layout, int64_t samples_before,
                         af::BlockStats *stats, const double *vad, hipStream_t stream, hipStream_t /*caller*/,
                         bool stats_cleared = false, const double *pre_power = nullptr, const int64_t *ready = nullptr) {
  // `stats_cleared`: the rows were zeroed (and partly filled) by an earlier kernel of this window: do not clear them again
  // `pre_power`: [block][stream] compressor-input block powers of this segment, left by the systolic EQ kernel that ran as
  // the window's pre-pass: an auto-makeup segment is then ONE launch
  // `ready`: the segment is a whole call whose input arrives window by window (LaunchArgs::ready); only the plain one-launch form
  // of the token-ring kernel follows such a counter -- the caller has checked that this is what the configuration takes
  const bool followed_counter = ready != nullptr;
  if (ready && (!e->extra_presets.empty() || e->kernel == AF_KERNEL_ROLES || roles_mode() != 0))
    return fail(AF_ERR_BACKEND, "internal: only the plain token-ring launch follows a ready counter");
  if (!e->extra_presets.empty())
    return launch_chain_multi(e, e->host_params.flags & ~run_in.flags, run_in.flags & af::kFlagInputDone, in, out, n_samples,
                              stream_stride, layout, samples_before, stats, stream, stats_cleared);
  af::ChainParams run = run_in;
  const int cb = run.control_block;
  const int64_t rows = ((n_samples + cb - 1) / cb) * e->n_streams;
  bool any_xf = false;
  for (int k = 0; k < run.n_eq_sections; ++k) any_xf |= run.eq[k].xf_remaining > 0;
  const bool ring_fits = af::ring_kernel_dynamic_lds(run.n_eq_sections, run.lim.lookahead_samples, any_xf) <= af::kMaxLdsBytes;
  const bool auto_makeup = (run.flags & af::kFlagCompressor) && run.comp.auto_makeup_enabled;
  const bool eq_first_deesser = (run.flags & af::kFlagDeesser) && (run.flags & af::kFlagEqBeforeDeesser);
  const bool quad_ok = !auto_makeup && !eq_first_deesser &&
                       af::quad_kernel_dynamic_lds(run.n_eq_sections, run.lim.lookahead_samples, any_xf) <= af::kMaxLdsBytes;
  int kernel = e->kernel;
  if (kernel == AF_KERNEL_AUTO || kernel == AF_KERNEL_ROLES) {
    // Kernel 2 (the token ring, 64 streams per workgroup) wherever its LDS fits: a launch lasts as long as ONE
    // workgroup needs for its streams' samples, whatever the batch, and since its waves carry priorities (in a serial
    // unit, and growing with the age of their chunk) that is shorter than kernel 3's (16 streams per workgroup, which
    // the same priorities slow down): 206-212 vs 217-225 ms for 4096 streams x 10 s of the dynamics chain, 209 vs 212 ms
    // for 256 streams.  Kernel 3 serves configurations whose limiter ring does not fit kernel 2's LDS layout.
    kernel = ring_fits ? AF_KERNEL_PHASED : (quad_ok ? AF_KERNEL_QUAD : AF_KERNEL_LANE_PER_STREAM);
  }
  if (kernel == AF_KERNEL_QUAD && !quad_ok)
    return fail(AF_ERR_UNSUPPORTED, "the quad kernel does not build auto-makeup or the EQ-before-de-esser order; use AF_KERNEL_PHASED");
  if (kernel == AF_KERNEL_PHASED && !ring_fits)
    return fail(AF_ERR_UNSUPPORTED, "the token-ring kernel needs more LDS than a CU has for this configuration");
  if (kernel == AF_KERNEL_LANE_PER_STREAM && af::lane_kernel_dynamic_lds(run.lim.lookahead_samples) > 90 * 1024)
    return fail(AF_ERR_UNSUPPORTED, "the lane-per-stream kernel cannot hold a limiter lookahead of %d samples in LDS",
                run.lim.lookahead_samples);
  if (auto_makeup && kernel != AF_KERNEL_PHASED)
    return fail(AF_ERR_UNSUPPORTED, "compressor auto-makeup is only built into the token-ring kernel");
  e->last_kernel_used = kernel;
  const bool deesser = (run.flags & af::kFlagDeesser) != 0;
  const bool eq_first = (run.flags & af::kFlagEqBeforeDeesser) != 0;
  const uint32_t front_flags = af::kFlagInputScrub | af::kFlagInputClamp | af::kFlagDcBlock | af::kFlagPreHighpass;
  const bool two_pass = (auto_makeup && !pre_power) || (deesser && eq_first);
  if (two_pass && kernel != AF_KERNEL_PHASED)
    return fail(AF_ERR_UNSUPPORTED, "EQ-before-de-esser order is only built around the token-ring kernel");
  // ---- the role pipeline (af_roles.hip): compressor and limiter as two kernels of dedicated serial waves + feed-forward
  // waves, LDS hand-over; the EQ and the block input statistics are the systolic EQ kernel's.  Where it serves the
  // configuration it replaces the token-ring launch (same state planes: the two can alternate mid-stream).
  {
    const int roles_env = roles_mode();
    const bool input_done = (run.flags & af::kFlagInputDone) != 0;
    const bool eq_pre_ok = layout == AF_LAYOUT_STREAM_MAJOR && !(run.flags & (af::kFlagDcBlock | af::kFlagPreHighpass)) &&
                           (!(run.flags & af::kFlagEq) || run.n_eq_sections <= 16);
    const bool comp_on = (run.flags & af::kFlagCompressor) != 0;
    const bool served = !two_pass && !deesser && !auto_makeup && af::lim_roles_serves(run) && (!comp_on || af::comp_roles_serves(run)) &&
                        (input_done || eq_pre_ok);
    if (served && (e->kernel == AF_KERNEL_ROLES || (e->kernel == AF_KERNEL_AUTO && roles_env != 0))) {
      // the limiter half on a stream of its own (the suppressor pipeline's: other CUs, a window behind the compressor)
      const hipStream_t lim_stream = (e->lim_stream && stream == e->aux_stream) ? e->lim_stream : stream;
      e->last_kernel_used = AF_KERNEL_ROLES;
      // AF_ROLES=2: the compressor stays on the token-ring kernel, which then ends at the compressor's output
      const bool ring_comp = comp_on && roles_env == 2 && ring_fits;
      af::ChainParams up = run;
      if (ring_comp) up.flags = (up.flags & ~af::kFlagLimiter) | af::kFlagCompOnly;
      if (!e->uploaded_valid || std::memcmp(&e->uploaded, &up, sizeof up) != 0) {
        e->uploaded = up;
        if (int rc = stage_upload(e, e->d_params, &e->uploaded, 1, stream)) return rc;
        e->uploaded_valid = true;
      }
      hipEvent_t t0 = nullptr, t1 = nullptr;
      if (e->timing) {
        AF_HIP(hipEventCreate(&t0));
        AF_HIP(hipEventCreate(&t1));
        AF_HIP(hipEventRecord(t0, stream));
      }
      if (!stats_cleared) AF_HIP(hipMemsetAsync(stats, 0, sizeof(af::BlockStats) * rows, stream));
      af::LaunchArgs ra{};
      ra.st64 = e->d_st64;
      ra.st32 = e->d_st32;
      ra.in = in;
      ra.out = out;
      ra.stats = stats;
      ra.status = e->d_status;
      ra.params = e->d_params;
      ra.n_samples = n_samples;
      ra.stream_stride = stream_stride;
      ra.samples_before = samples_before;
      ra.n_streams = e->n_streams;
      ra.layout = layout;
      if (!input_done) {  // EQ (or a plain pass when it is off) + block input statistics
        AF_HIP(af::launch_eq_systolic(e->d_params, nullptr, e->d_st64, in, out, nullptr, nullptr, 0, 0, stats, any_xf, n_samples,
                                      stream_stride, e->n_streams, stream));
        ra.in = out;
        e->last_launches += 1;
      }
      if (ring_comp) {
        af::ChainParams shape = up;
        shape.flags = (shape.flags & ~af::kFlagEq) | af::kFlagInputDone;  // (the systolic EQ kernel above did both)
        if (std::memcmp(&e->uploaded, &shape, sizeof shape) != 0) {
          e->uploaded = shape;
          if (int rc = stage_upload(e, e->d_params, &e->uploaded, 1, stream)) return rc;
        }
        AF_HIP(af::launch_chain_ring(ra, shape.n_eq_sections, shape.lim.lookahead_samples, any_xf, e->ring_variant, false, stream));
        ra.in = out;
        e->last_launches += 1;
      } else if (comp_on) {
        AF_HIP(af::launch_chain_comp_roles(ra, run.comp.sidechain_highpass_enabled != 0, run.comp.adaptive_release != 0, stream));
        ra.in = out;
        e->last_launches += 1;
      }
      if (e->timing) {  // (with the limiter on its own stream the bracket holds what the chain stream did)
        AF_HIP(hipEventRecord(t1, stream));
        e->chain_ms_events.push_back({t0, t1});
      }
      if (lim_stream != stream) {
        hipEvent_t comp_done;
        if (int rc = engine_event(e, &comp_done)) return rc;
        AF_HIP(hipEventRecord(comp_done, stream));
        AF_HIP(hipStreamWaitEvent(lim_stream, comp_done, 0));
      }
      AF_HIP(af::launch_chain_lim_roles(ra, run.lim.lookahead_samples, lim_stream));
      e->last_launches += 1;
      advance_crossfades(e, n_samples);
      return AF_OK;
    }
    if (e->kernel == AF_KERNEL_ROLES) kernel = ring_fits ? AF_KERNEL_PHASED : (quad_ok ? AF_KERNEL_QUAD : AF_KERNEL_LANE_PER_STREAM);  // not served: as AUTO
  }
  af::LaunchArgs a{};
  a.st64 = e->d_st64;
  a.st32 = e->d_st32;
  a.in = in;
  a.out = out;
  a.stats = stats;
  a.status = e->d_status;
  a.params = e->d_params;
  a.n_samples = n_samples;
  a.stream_stride = stream_stride;
  a.samples_before = samples_before;
  a.n_streams = e->n_streams;
  a.layout = layout;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  if (e->timing) {
    AF_HIP(hipEventCreate(&t0));
    AF_HIP(hipEventCreate(&t1));
    AF_HIP(hipEventRecord(t0, stream));
  }
  if ((two_pass || deesser) && rows > e->stats_pre_capacity) {  // (per-call scratch: the old rows are retired, not freed)
    int64_t cap_pre = e->stats_pre_capacity * (int64_t)sizeof(af::BlockStats), cap_de = cap_pre;
    if (int rc = grow_device(e, reinterpret_cast<void **>(&e->d_stats_pre), &cap_pre, rows * (int64_t)sizeof(af::BlockStats), stream)) return rc;
    if (int rc = grow_device(e, reinterpret_cast<void **>(&e->d_stats_de), &cap_de, rows * (int64_t)sizeof(af::BlockStats), stream)) return rc;
    e->stats_pre_capacity = std::min(cap_pre, cap_de) / (int64_t)sizeof(af::BlockStats);
  }
  const af::BlockStats *input_rows = nullptr;  // where the block input statistics end up when a side pass saw the input
  if (deesser) {
    // the de-esser pass reads the unmodified parameter block (its own copy: the chain kernels get edited flags)
    if (!e->uploaded_de_valid || std::memcmp(&e->uploaded_de, &run_in, sizeof run_in) != 0) {
      e->uploaded_de = run_in;
      if (int rc = stage_upload(e, e->d_params_de, &e->uploaded_de, 1, stream)) return rc;
      e->uploaded_de_valid = true;
    }
    AF_HIP(hipMemsetAsync(e->d_stats_de, 0, sizeof(af::BlockStats) * rows, stream));
    if (!eq_first) {
      // routing.rs: pre-filter -> de-esser -> EQ ...: this pass takes the front end with it
      AF_HIP(af::launch_deesser(e->d_params_de, e->d_st64, e->d_st32, in, out, e->d_stats_de, n_samples, stream_stride,
                                e->n_streams, layout, true, false, stream));
      e->last_launches += 1;
      a.in = out;
      run.flags &= ~front_flags;
      input_rows = e->d_stats_de;
    }
    run.flags &= ~af::kFlagDeesser;
  }
  if (kernel == AF_KERNEL_PHASED) {
    // the ring kernel writes each stats field from the token that owns it; untouched fields must read 0
    if (!stats_cleared) AF_HIP(hipMemsetAsync(stats, 0, sizeof(af::BlockStats) * rows, stream));
    if (two_pass) {
      af::ChainParams pre = run, post = run;
      pre.flags = (pre.flags & ~(af::kFlagCompressor | af::kFlagLimiter)) | af::kFlagPrePass;
      post.flags &= ~(af::kFlagEq | front_flags);
      // both variants are uploaded only when they change (first launch; while a coefficient crossfade advances), through
      // pinned slots: no host wait inside a window loop
      if (!e->uploaded_pre_valid || std::memcmp(&e->uploaded_pre, &pre, sizeof pre) != 0) {
        e->uploaded_pre = pre;
        if (int rc = stage_upload(e, e->d_params_pre, &e->uploaded_pre, 1, stream)) return rc;
        e->uploaded_pre_valid = true;
      }
      if (!e->uploaded_valid || std::memcmp(&e->uploaded, &post, sizeof post) != 0) {
        e->uploaded = post;
        if (int rc = stage_upload(e, e->d_params, &e->uploaded, 1, stream)) return rc;
        e->uploaded_valid = true;
      }
      AF_HIP(hipMemsetAsync(e->d_stats_pre, 0, sizeof(af::BlockStats) * rows, stream));
      af::LaunchArgs a1 = a;
      a1.params = e->d_params_pre;
      a1.stats = e->d_stats_pre;
      AF_HIP(af::launch_chain_ring(a1, pre.n_eq_sections, pre.lim.lookahead_samples, any_xf, e->ring_variant, false, stream));
      if (!input_rows) input_rows = e->d_stats_pre;
      const af::BlockStats *power_rows = e->d_stats_pre;
      if (deesser && eq_first) {
        // ... -> EQ -> de-esser -> compressor: the compressor-input block power is the de-esser's output power
        AF_HIP(af::launch_deesser(e->d_params_de, e->d_st64, e->d_st32, out, out, e->d_stats_de, n_samples, stream_stride,
                                  e->n_streams, layout, false, auto_makeup, stream));
        e->last_launches += 1;
        power_rows = e->d_stats_de;
      }
      af::LaunchArgs a2 = a;
      a2.in = out;
      if (auto_makeup) {
        a2.pre_stats = power_rows;
        a2.vad_prob = vad;
      }
      AF_HIP(af::launch_chain_ring(a2, post.n_eq_sections, post.lim.lookahead_samples, any_xf, e->ring_variant, auto_makeup, stream));
      e->last_launches += 2;
    } else {
      if (!e->uploaded_valid || std::memcmp(&e->uploaded, &run, sizeof run) != 0) {
        e->uploaded = run;
        if (int rc = stage_upload(e, e->d_params, &e->uploaded, 1, stream)) return rc;
        e->uploaded_valid = true;
      }
      if (auto_makeup) {  // the systolic EQ kernel was this segment's pre-pass
        a.pre_power = pre_power;
        a.vad_prob = vad;
      }
      a.ready = ready;
      ready = nullptr;  // (taken)
      AF_HIP(af::launch_chain_ring(a, run.n_eq_sections, run.lim.lookahead_samples, any_xf, e->ring_variant, auto_makeup, stream));
      e->last_launches += 1;
    }
  } else {
    if (!e->uploaded_valid || std::memcmp(&e->uploaded, &run, sizeof run) != 0) {
      e->uploaded = run;
      if (int rc = stage_upload(e, e->d_params, &e->uploaded, 1, stream)) return rc;
      e->uploaded_valid = true;
    }
    if (kernel == AF_KERNEL_QUAD) {
      AF_HIP(hipMemsetAsync(stats, 0, sizeof(af::BlockStats) * rows, stream));  // fields are written by their tokens
      AF_HIP(af::launch_chain_quad(a, run.n_eq_sections, run.lim.lookahead_samples, any_xf, e->ring_variant ? e->ring_variant / 100 : 12, stream));
    } else {
      AF_HIP(af::launch_chain_lane(a, run.lim.lookahead_samples, stream));
    }
    e->last_launches += 1;
  }
  if (ready) return fail(AF_ERR_BACKEND, "internal: a launch that follows a ready counter took a path that does not read it");
  if (input_rows || deesser) {
    AF_HIP(af::launch_merge_side_stats(stats, input_rows, deesser ? e->d_stats_de : nullptr, rows, stream));
    e->last_launches += 1;
  }
  if (e->timing) {
    AF_HIP(hipEventRecord(t1, stream));
    e->chain_ms_events.push_back({t0, t1});
  }
  if (!followed_counter) advance_crossfades(e, n_samples);  // (a one-launch call: the caller moves the counters window by window)
  return AF_OK;
}


// an event of the engine's pool, valid until the end of the current process call
int engine_event(af_engine *e, hipEvent_t *out_ev) {
  if (e->ev_cursor == e->sync_events.size()) {
    hipEvent_t ev;
    AF_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    e->sync_events.push_back(ev);
  }
  *out_ev = e->sync_events[e->ev_cursor++];
  return AF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The stage-pipeline form of the chain (af_stages.hip).  Which configurations it serves:
bool stage_pipe_serves_one(const af::ChainParams &run) {
  if (run.flags & (af::kFlagDcBlock | af::kFlagPreHighpass | af::kFlagPrePass)) return false;
  // the de-esser in its default place (ahead of the EQ) runs as stages of its own; EQ-before-de-esser stays on kernel 2
  if ((run.flags & af::kFlagDeesser) && (run.flags & af::kFlagEqBeforeDeesser)) return false;
  if ((run.flags & af::kFlagEq) && run.n_eq_sections > 16) return false;
  if ((run.flags & af::kFlagLimiter) && run.lim.lookahead_samples > af::kMaxLookahead) return false;
  return true;
}
// `run`: preset 0 as the chain will see it (the front end's flags stripped when the suppressor's pre-pass owns them)
bool stage_pipe_serves(af_engine *e, const af::ChainParams &run, int32_t layout) {
  if (layout != AF_LAYOUT_STREAM_MAJOR || !stage_pipe_serves_one(run)) return false;
  const uint32_t strip = e->host_params.flags & ~run.flags;
  for (int k = 1; k <= (int)e->extra_presets.size(); ++k) {
    // several presets: every group's stages run as roles of the same dispatches, so the presets must agree on which stages
    // there are and which code paths they take (what differs freely: every coefficient, the EQ, the limiter's lookahead)
    af::ChainParams other = preset_params(e, k);
    other.flags &= ~strip;
    const uint32_t shape = af::kFlagCompressor | af::kFlagLimiter | af::kFlagDeesser;
    if (!stage_pipe_serves_one(other) || (other.flags & shape) != (run.flags & shape) ||
        other.comp.sidechain_highpass_enabled != run.comp.sidechain_highpass_enabled ||
        other.comp.adaptive_release != run.comp.adaptive_release || other.comp.auto_makeup_enabled != run.comp.auto_makeup_enabled ||
        other.control_block != run.control_block)
      return false;
  }
  return true;
}

size_t pow2_at_least(int64_t n) {
  size_t p = 1;
  while ((int64_t)p < n) p <<= 1;
  return p;
}

// rings sized for windows of up to `tw_max` samples
int stage_pipe_prepare(af_engine *e, int64_t tw_max) {
  auto &sp = e->pipe;
  {  // the per-block arrays (makeup gains, block powers): a reset may have brought a shorter control block, i.e. more blocks
    const int cb = e->host_params.control_block;
    const int64_t rows = ((std::max(tw_max, sp.tw_max) + cb - 1) / cb + 1) * e->n_streams;
    if (rows > sp.mk_rows) {
      if (sp.d_mk || sp.d_bp) AF_HIP(hipDeviceSynchronize());
      if (sp.d_mk) (void)hipFree(sp.d_mk);
      if (sp.d_bp) (void)hipFree(sp.d_bp);
      sp.d_mk = sp.d_bp = nullptr;
      AF_HIP(hipMalloc(&sp.d_mk, sizeof(double) * rows * af_engine::StagePipe::kMkSets));
      AF_HIP(hipMalloc(&sp.d_bp, sizeof(double) * rows * af_engine::StagePipe::kBpSets));
      sp.mk_rows = rows;
    }
  }
  if (sp.rings.xe && tw_max <= sp.tw_max && sp.with_deesser == ((e->host_params.flags & af::kFlagDeesser) != 0)) return AF_OK;
  if (sp.rings.xe) {  // grow: only between calls of a fresh engine (the rings hold the histories)
    if (sp.windows > 0 && tw_max > sp.tw_max)
      return fail(AF_ERR_UNSUPPORTED, "a call of %lld samples per window after smaller ones: the stage pipeline's rings were sized for %lld",
                  (long long)tw_max, (long long)sp.tw_max);
    AF_HIP(hipDeviceSynchronize());
    for (void *p : sp.allocs) (void)hipFree(p);
    sp.allocs.clear();
    sp.rings = af::StageRings{};
  }
  const int64_t groups = (e->n_streams + 63) / 64;
  const int64_t hist = 2 * (af::kMaxLookahead + 1) + 64;
  // a ring holds the windows between its producer and its last consumer, one more, and the history: the stages advance in lock
  // step, an f64 ring's reader two windows behind its writer at most, the EQ output's last reader seven
  const bool deesser = (e->host_params.flags & af::kFlagDeesser) != 0;
  // (with the de-esser: a band's coefficients wait three launch steps for the third dynamic EQ of the cascade, and the chain
  // input nine for nothing -- the EQ now reads the de-esser's output ring)
  const size_t r64 = pow2_at_least((deesser ? 6 : 4) * tw_max + hist), r32 = pow2_at_least(10 * tw_max + hist);
  sp.rings.rows_f64 = (int32_t)r64;
  sp.rings.rows_f32 = (int32_t)r32;
  auto ring32 = [&](float **p) -> hipError_t {
    hipError_t err = hipMalloc(p, sizeof(float) * r32 * 64 * groups);
    if (err != hipSuccess) return err;
    sp.allocs.push_back(*p);
    return hipMemset(*p, 0, sizeof(float) * r32 * 64 * groups);
  };
  auto ring64 = [&](double **p) -> hipError_t {
    hipError_t err = hipMalloc(p, sizeof(double) * r64 * 64 * groups);
    if (err != hipSuccess) return err;
    sp.allocs.push_back(*p);
    return hipMemset(*p, 0, sizeof(double) * r64 * 64 * groups);
  };
  af::StageRings &r = sp.rings;
  for (float **p : {&r.xi, &r.xe, &r.xc, &r.sfx, &r.xl, &r.itp, &r.tgt, &r.gt, &r.od}) AF_HIP(ring32(p));
  for (double **p : {&r.d, &r.pr, &r.low_e, &r.voiced_e, &r.pres_e, &r.rms_e, &r.ipk_db, &r.rms_db, &r.w_db, &r.peak_db, &r.target, &r.gr, &r.glin, &r.fast_r, &r.slow_r, &r.tgt_ms, &r.tg, &r.g})
    AF_HIP(ring64(p));
  if (deesser) {
    for (int b = 0; b < 3; ++b) {
      for (double **p : {&r.de_env[b], &r.de_ct[b], &r.de_ratio[b], &r.de_aux[b], &r.de_tr[b], &r.de_gdb[b], &r.de_red[b]}) AF_HIP(ring64(p));
      AF_HIP(ring32(&r.de_upd[b]));
      for (int j = 0; j < 5; ++j) AF_HIP(ring64(&r.de_c[b][j]));
      AF_HIP(ring32(&r.de_y[b]));
    }
    AF_HIP(ring64(&r.de_bb));
  }
  sp.with_deesser = deesser;
  sp.tw_max = tw_max;
  if (!sp.stream) {  // (a queue of its own: a CU-masked stream with every CU enabled; plain streams share a few hardware queues)
    hipDeviceProp_t prop;
    AF_HIP(hipGetDeviceProperties(&prop, e->device));
    std::vector<uint32_t> mask((size_t)(prop.multiProcessorCount + 31) / 32, 0u);
    for (int bit = 0; bit < prop.multiProcessorCount; ++bit) mask[bit >> 5] |= 1u << (bit & 31);
    if (hipExtStreamCreateWithCUMask(&sp.stream, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
      (void)hipGetLastError();
      AF_HIP(hipStreamCreateWithFlags(&sp.stream, hipStreamNonBlocking));
    }
  }
  return AF_OK;
}

int stage_pipe_clear(af_engine *e) {  // a fresh engine: the histories are zeros
  auto &sp = e->pipe;
  if (!sp.rings.xe) return AF_OK;
  const int64_t groups = (e->n_streams + 63) / 64;
  af::StageRings &r = sp.rings;
  for (float *p : {r.xi, r.xe, r.xc, r.sfx, r.xl, r.itp, r.tgt, r.gt, r.od}) AF_HIP(hipMemset(p, 0, sizeof(float) * r.rows_f32 * 64 * groups));
  for (double *p : {r.d, r.pr, r.low_e, r.voiced_e, r.pres_e, r.rms_e, r.ipk_db, r.rms_db, r.w_db, r.peak_db, r.target, r.gr, r.glin, r.fast_r, r.slow_r, r.tgt_ms, r.tg, r.g})
    AF_HIP(hipMemset(p, 0, sizeof(double) * r.rows_f64 * 64 * groups));
  if (sp.with_deesser) {
    for (int b = 0; b < 3; ++b) {
      for (double *p : {r.de_env[b], r.de_ct[b], r.de_ratio[b], r.de_aux[b], r.de_tr[b], r.de_gdb[b], r.de_red[b]}) AF_HIP(hipMemset(p, 0, sizeof(double) * r.rows_f64 * 64 * groups));
      AF_HIP(hipMemset(r.de_upd[b], 0, sizeof(float) * r.rows_f32 * 64 * groups));
      for (int j = 0; j < 5; ++j) AF_HIP(hipMemset(r.de_c[b][j], 0, sizeof(double) * r.rows_f64 * 64 * groups));
      AF_HIP(hipMemset(r.de_y[b], 0, sizeof(float) * r.rows_f32 * 64 * groups));
    }
    AF_HIP(hipMemset(r.de_bb, 0, sizeof(double) * r.rows_f64 * 64 * groups));
  }
  sp.windows = 0;
  return AF_OK;
}

// The parameter block(s) the stage kernels read: preset 0 alone, or all presets as an array beside the group -> preset table.
int stage_chain_params(af_engine *e, hipStream_t stream) {
  auto &sp = e->pipe;
  const int n_presets = 1 + (int)e->extra_presets.size();
  std::vector<af::ChainParams> runs((size_t)n_presets);
  sp.w_min = 1 << 30;
  for (int k = 0; k < n_presets; ++k) {
    runs[k] = preset_params(e, k);
    runs[k].flags &= ~sp.strip;
    sp.w_min = std::min<int32_t>(sp.w_min, runs[k].lim.lookahead_samples + 1);
  }
  if (n_presets == 1) {
    if (!e->uploaded_valid || std::memcmp(&e->uploaded, &runs[0], sizeof runs[0]) != 0) {
      e->uploaded = runs[0];
      if (int rc = stage_upload(e, e->d_params, &e->uploaded, 1, stream)) return rc;
      e->uploaded_valid = true;
    }
    sp.d_chain = e->d_params;
  } else {
    if (e->uploaded_multi.size() != runs.size() ||
        std::memcmp(e->uploaded_multi.data(), runs.data(), sizeof(af::ChainParams) * runs.size()) != 0) {
      e->uploaded_multi = runs;
      if (int rc = stage_upload(e, e->d_params_multi, runs.data(), runs.size(), stream)) return rc;
    }
    sp.d_chain = e->d_params_multi;
  }
  return AF_OK;
}

// ---- the pipeline as one launch per step (af_stages.h, DiagArgs): launch j runs stage k on window j - skew(k) -----------------
// The stages of this configuration in chain order, each one launch step behind the stage it reads from.
struct StagePlan {
  int stage[af::kStCount], skew[af::kStCount], n = 0;
  int depth = 0;  // the last stage's skew: launches a window needs to leave the pipeline after it entered
};
StagePlan stage_plan(const af::ChainParams &run) {
  StagePlan p;
  const bool comp = (run.flags & af::kFlagCompressor) != 0, lim = (run.flags & af::kFlagLimiter) != 0;
  auto add = [&](int k, int sk) { p.stage[p.n] = k; p.skew[p.n] = sk; ++p.n; p.depth = std::max(p.depth, sk); return sk; };
  int at = 0;
  if (run.flags & af::kFlagDeesser) {
    // the de-esser ahead of the EQ (deesser.rs:405-547): transposing loader | three detectors | levels and confidence targets |
    // three confidence / baseline recurrences | target scaling | three reduction smoothers with the gain hold | coefficients
    // (and the block's figure) | three cascaded dynamic EQs
    add(af::kStDe0, 0);
    for (int k : {af::kStDe1a, af::kStDe1b, af::kStDe1c}) add(k, 1);
    add(af::kStDe2, 2);
    for (int k : {af::kStDe3a, af::kStDe3b, af::kStDe3c}) add(k, 3);
    add(af::kStDe4s, 4);
    for (int k : {af::kStDe4a, af::kStDe4b, af::kStDe4c}) add(k, 5);
    add(af::kStDe4t, 6);
    add(af::kStDe5, 6);
    add(af::kStDe6a, 7);
    add(af::kStDe6b, 8);
    add(af::kStDe6c, 9);
    at = add(af::kStEq, 10);
  } else {
    at = add(af::kStEq, 0);
  }
  const int eq_at = at;
  add(af::kStIn, 1);  // (the xi ring: written by the EQ stage, or by the de-esser's loader, one step earlier)
  if (comp) {
    for (int k : {af::kStCompA, af::kStCompA2, af::kStF1, af::kStCompC, af::kStF2, af::kStCompE}) at = add(k, at + 1);
    if (run.comp.adaptive_release) {
      add(af::kStFR, at + 1);
      add(af::kStRel, at + 2);
    }
    if (run.comp.auto_makeup_enabled) {
      add(af::kStPow, eq_at + 1);
      at = add(af::kStF3a, at + 1);
      at = add(af::kStMakeup, at + 1);
    } else {
      at = add(af::kStF3, at + 1);
    }
  }
  if (lim)
    for (int k : {af::kStF4, af::kStLim, af::kStF5, af::kStTp}) at = add(k, at + 1);
  at = add(af::kStOut, at + 1);
  add(af::kStF6, at + 1);
  return p;
}

// launch step j of a call whose windows are `wins`: every stage whose window exists
int stage_diag_step(af_engine *e, const af::ChainParams &run, const StagePlan &plan, const std::vector<af::DiagWin> &wins, int64_t j,
                    hipStream_t stream) {
  auto &sp = e->pipe;
  af::DiagArgs d{};
  d.base.params = sp.d_chain;
  d.base.group_preset = e->extra_presets.empty() ? nullptr : e->d_group_preset;
  d.base.st64 = e->d_st64;
  d.base.st32 = e->d_st32;
  d.base.n_streams = e->n_streams;
  d.base.w_min = sp.w_min;
  d.base.r = sp.rings;
  d.params_eq = e->d_params_eq;
  d.flags = run.flags;
  d.sidechain = run.comp.sidechain_highpass_enabled;
  d.adaptive = run.comp.adaptive_release;
  d.auto_makeup = (run.flags & af::kFlagCompressor) && run.comp.auto_makeup_enabled;
  d.base.stream_stride = sp.call_stride;
  d.deesser = (run.flags & af::kFlagDeesser) ? 1 : 0;
  static const int debug_skip = [] {  // AF_STAGE_SKIP=<StageId>: timing probe (that stage does nothing; results are garbage)
    const char *env = std::getenv("AF_STAGE_SKIP");
    return env ? std::atoi(env) : -1;
  }();
  d.debug_skip = debug_skip;
  // two dispatches per step: the one-wave workgroups (serial stages and F4), then the wide stages; with the de-esser a third
  // for its serial stages
  for (int pass = 0; pass < (d.deesser ? 3 : 2); ++pass) {
    unsigned blocks = 0;
    d.n_roles = 0;
    for (int i = 0; i < plan.n; ++i) {
      const int k = plan.stage[i];
      if (af::stage_dispatch_kind(k) != pass) continue;
      const int64_t wi = j - plan.skew[i];
      if (wi < 0 || wi >= (int64_t)wins.size()) continue;
      af::DiagRole &role = d.roles[d.n_roles++];
      role.stage = k;
      role.win = wins[(size_t)wi];
      unsigned gy = 1;
      role.gx = af::stage_role_blocks(k, role.win.n0, role.win.n, e->n_streams, d.base.w_min, &gy);
      role.first_block = blocks;
      blocks += role.gx * gy;
    }
    if (d.n_roles == 0) continue;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (e->timing && pass == 0) {  // the serial stages' dispatch is what a step lasts: what af_engine_last_chain_launch_ms reports
      AF_HIP(hipEventCreate(&t0));
      AF_HIP(hipEventCreate(&t1));
      AF_HIP(hipEventRecord(t0, stream));
    }
    AF_HIP(af::launch_stage_diag(d, blocks, pass, stream));
    if (t0) {
      AF_HIP(hipEventRecord(t1, stream));
      e->chain_ms_events.push_back({t0, t1});
    }
    e->last_launches += 1;
  }
  return AF_OK;
}

bool rnn_stream_wanted() {
  static const bool on = [] {
    const char *env = std::getenv("AF_RNN_STREAM");
    return env && std::atoi(env) == 1;
  }();
  return on;
}

// The engine's side streams (created once).  With queue CU masks: the chain stream on as many CUs as the chain has workgroups,
// every other stream on the rest.
int ensure_side_streams(af_engine *e, hipStream_t stream) {
  if (std::getenv("AF_SERIAL_STREAMS")) {  // diagnostic: every stage on the caller's stream (per-kernel times without overlap)
    e->aux_stream = e->pre_stream = e->ana_stream = e->fin_stream = e->eq_stream = stream;
    e->borrowed_streams = true;
  }

  if (!e->aux_stream) {
    // CU partition.  A 16-wave chain workgroup needs a whole CU (it fills the register file), and the suppressor's
    // kernels keep thousands of small, some of them long-lived, workgroups in flight: left to the dispatcher, every chain
    // launch waits for CUs to drain and runs beside strangers.  So the chain stream is confined to as many CUs as it has
    // workgroups (mask bits 0.. select the same CU indices on every XCD: tools/probe/cu_mask_probe.hip) and the
    // suppressor's streams to the rest; neither side ever waits for the other's workgroups to leave.
    int chain_cus = 0;
    const char *env = std::getenv("AF_CU_PARTITION");
    const int chain_groups = (e->n_streams + 63) / 64;
    hipDeviceProp_t prop;
    AF_HIP(hipGetDeviceProperties(&prop, e->device));
    const int total_cus = prop.multiProcessorCount;
    if (!(env && std::atoi(env) == 0) && total_cus % 32 == 0 && total_cus <= 1024) {
      // (a power of two: the workgroups of a launch are dealt to the XCDs in turn and 48 or 56 enabled CUs leave some of them
      // with two workgroups each -- 3072 streams: 356 ms of chain launches per step on 48 CUs, 197 on 64)
      int pow2 = 8;
      while (pow2 < chain_groups) pow2 *= 2;
      chain_cus = env && std::atoi(env) > 0 ? std::atoi(env) : pow2;
      if (chain_cus * 2 > total_cus) chain_cus = 0;  // a chain that wants half the chip or more shares all of it
    }
    if (chain_cus > 0) {
      // AF_ROLES=2: the limiter half of the chain gets CUs of its own (AF_LIM_CUS, default as many as the chain), taken from
      // the suppressor's share
      int lim_cus = 0;
      if (roles_mode() == 2) {
        const char *lenv = std::getenv("AF_LIM_CUS");
        lim_cus = lenv ? std::atoi(lenv) : chain_cus;
        if (lim_cus < 0 || chain_cus + lim_cus + 32 > total_cus) lim_cus = 0;
      }
      std::vector<uint32_t> chain_mask(total_cus / 32, 0u), rest_mask(total_cus / 32, 0u), lim_mask(total_cus / 32, 0u);
      // AF_CU_PATTERN (placement probe): 0 = the chain takes mask bits 0.. (CU indices 0.. of every XCD); 1 = every other CU
      // index (bit / 8 even); 2 = the highest bits
      static const int pattern = [] { const char *v = std::getenv("AF_CU_PATTERN"); return v ? std::atoi(v) : 0; }();
      for (int bit = 0; bit < total_cus; ++bit) {
        int rank = bit;  // the chain takes ranks 0 .. chain_cus-1
        if (pattern == 1) rank = ((bit / 8) % 2 == 0) ? (bit / 16) * 8 + bit % 8 : total_cus / 2 + (bit / 16) * 8 + bit % 8;
        else if (pattern == 2) rank = total_cus - 1 - bit;
        (rank < chain_cus ? chain_mask : (rank < chain_cus + lim_cus ? lim_mask : rest_mask))[bit >> 5] |= 1u << (bit & 31);
      }
      hipError_t err = hipExtStreamCreateWithCUMask(&e->aux_stream, (uint32_t)chain_mask.size(), chain_mask.data());
      if (err == hipSuccess && lim_cus > 0) err = hipExtStreamCreateWithCUMask(&e->lim_stream, (uint32_t)lim_mask.size(), lim_mask.data());
      if (err == hipSuccess) err = hipExtStreamCreateWithCUMask(&e->pre_stream, (uint32_t)rest_mask.size(), rest_mask.data());
      if (err == hipSuccess) err = hipExtStreamCreateWithCUMask(&e->ana_stream, (uint32_t)rest_mask.size(), rest_mask.data());
      if (err == hipSuccess) err = hipExtStreamCreateWithCUMask(&e->syn_stream, (uint32_t)rest_mask.size(), rest_mask.data());
      if (err == hipSuccess) err = hipExtStreamCreateWithCUMask(&e->fin_stream, (uint32_t)rest_mask.size(), rest_mask.data());
      if (err == hipSuccess && rnn_stream_wanted()) err = hipExtStreamCreateWithCUMask(&e->rnn_stream, (uint32_t)rest_mask.size(), rest_mask.data());
      if (err == hipSuccess) err = hipExtStreamCreateWithCUMask(&e->eq_stream, (uint32_t)rest_mask.size(), rest_mask.data());
      if (err != hipSuccess) {  // platform without queue CU masks: plain streams
        (void)hipGetLastError();
        for (hipStream_t *sp : {&e->aux_stream, &e->pre_stream, &e->ana_stream, &e->syn_stream, &e->fin_stream, &e->eq_stream, &e->lim_stream,
                                &e->rnn_stream}) {
          if (*sp) (void)hipStreamDestroy(*sp);
          *sp = nullptr;
        }
        chain_cus = 0;
      }
    }
    e->partition_chain_cus = chain_cus;
  }
  if (!e->aux_stream) AF_HIP(hipStreamCreateWithFlags(&e->aux_stream, hipStreamNonBlocking));
  if (!e->pre_stream) AF_HIP(hipStreamCreateWithFlags(&e->pre_stream, hipStreamNonBlocking));
  if (!e->ana_stream) AF_HIP(hipStreamCreateWithFlags(&e->ana_stream, hipStreamNonBlocking));
  if (!e->fin_stream) AF_HIP(hipStreamCreateWithFlags(&e->fin_stream, hipStreamNonBlocking));
  if (!e->rnn_stream && !e->borrowed_streams && rnn_stream_wanted()) AF_HIP(hipStreamCreateWithFlags(&e->rnn_stream, hipStreamNonBlocking));
  if (!e->eq_stream) AF_HIP(hipStreamCreateWithFlags(&e->eq_stream, hipStreamNonBlocking));
  return AF_OK;
}

// the EQ stage's parameter block for the window about to enter the pipeline (stream-ordered behind the previous window's launch)
int stage_diag_eq_params(af_engine *e, hipStream_t stream, bool *crossfade, int32_t *slot_out) {
  const int n_presets = 1 + (int)e->extra_presets.size();
  std::vector<af::ChainParams> runs((size_t)n_presets);
  *crossfade = false;
  bool deesser = false;
  for (int p = 0; p < n_presets; ++p) {
    runs[p] = preset_params(e, p);
    runs[p].flags &= ~e->pipe.strip;
    for (int k = 0; k < runs[p].n_eq_sections; ++k) *crossfade |= runs[p].eq[k].xf_remaining > 0;
    if (runs[p].flags & af::kFlagDeesser) {
      deesser = true;
      runs[p].flags &= ~(af::kFlagInputScrub | af::kFlagInputClamp);  // the de-esser's loader stage scrubbed the input already
    }
  }
  // With the de-esser every window keeps a parameter block of its own for as long as it is in the pipeline: the de-esser's
  // stages read their filters' crossfade counters as of the window's first sample up to eight launch steps after it entered.
  constexpr int kSlots = kEqParamSlots;
  if (!e->d_params_eq || e->eq_params_presets != n_presets) {
    if (e->d_params_eq) AF_HIP(hipFree(e->d_params_eq));
    e->d_params_eq = nullptr;
    AF_HIP(hipMalloc(&e->d_params_eq, sizeof(af::ChainParams) * n_presets * kSlots));
    e->eq_params_presets = n_presets;
    e->uploaded_eq.clear();
  }
  if (deesser) {
    const int slot = (int)(e->eq_slot_cursor++ % kSlots);
    if (int rc = stage_upload(e, e->d_params_eq + (size_t)slot * n_presets, runs.data(), runs.size(), stream)) return rc;
    e->uploaded_eq.clear();
    *slot_out = slot * n_presets;
    return AF_OK;
  }
  *slot_out = 0;
  if (e->uploaded_eq.size() != runs.size() || std::memcmp(e->uploaded_eq.data(), runs.data(), sizeof(af::ChainParams) * runs.size()) != 0) {
    e->uploaded_eq = runs;
    if (int rc = stage_upload(e, e->d_params_eq, runs.data(), runs.size(), stream)) return rc;
  }
  return AF_OK;
}

}  // namespace

extern "C" {

int af_version(void) { return 100; }
const char *af_last_error(void) { return g_last_error.c_str(); }

int af_device_count(void) {
  int n = 0;
  hipError_t err = hipGetDeviceCount(&n);
  if (err != hipSuccess) return fail(AF_ERR_BACKEND, "hipGetDeviceCount failed: %s", hipGetErrorString(err));
  return n;
}

int af_engine_create(double sample_rate, int32_t n_streams, int32_t device, af_engine **out) {
  if (!out) return fail(AF_ERR_INVALID_ARGUMENT, "out is null");
  *out = nullptr;
  if (!std::isfinite(sample_rate) || sample_rate <= 0.0)
    return fail(AF_ERR_INVALID_ARGUMENT, "sample_rate must be positive and finite");
  if (n_streams <= 0) return fail(AF_ERR_INVALID_ARGUMENT, "n_streams must be positive");
  if (device < 0) return fail(AF_ERR_INVALID_ARGUMENT, "device must be >= 0");
  *out = new af_engine(sample_rate, n_streams, device);
  return AF_OK;
}

void af_engine_destroy(af_engine *e) {
  if (!e) return;
  if (e->d_params || e->d_st64 || e->d_stats || e->d_io || e->d_pending || e->d_asm || e->d_trace || e->d_group_preset || e->d_params_multi) {
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(e->d_params);
    (void)hipFree(e->d_st64);
    (void)hipFree(e->d_st32);
    (void)hipFree(e->d_stats);
    (void)hipFree(e->d_stats_pre);
    (void)hipFree(e->d_params_pre);
    (void)hipFree(e->d_params_de);
    (void)hipFree(e->d_stats_de);
    (void)hipFree(e->d_vad);
    (void)hipFree(e->d_status);
    (void)hipFree(e->d_ready);
    (void)hipFree(e->d_io);
    (void)hipFree(e->d_pending);
    (void)hipFree(e->d_group_preset);
    (void)hipFree(e->d_params_eq);
    (void)hipFree(e->d_params_multi);
    (void)hipFree(e->d_asm);
    (void)hipFree(e->d_trace);
    (void)hipFree(e->d_block_power);
    if (e->ev_start) (void)hipEventDestroy(e->ev_start);
    if (e->ev_stop) (void)hipEventDestroy(e->ev_stop);
    if (e->ev_mid) (void)hipEventDestroy(e->ev_mid);
  }
  if (e->pipe.rings.xe || e->pipe.d_mk) {
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    for (void *p : e->pipe.allocs) (void)hipFree(p);
    (void)hipFree(e->pipe.d_mk);
    (void)hipFree(e->pipe.d_bp);
  }
  if (!e->retired.empty() || e->stager.pinned) {
    (void)hipSetDevice(e->device);
    (void)collect_retired(e, true);
    for (hipEvent_t ev : e->stager.done)
      if (ev) { (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); }
    if (e->stager.pinned) (void)hipHostFree(e->stager.pinned);
  }
  if (e->pipe.stream) (void)hipStreamDestroy(e->pipe.stream);
  for (hipEvent_t ev : e->sync_events) (void)hipEventDestroy(ev);
  for (auto &pr : e->chain_ms_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  if (e->borrowed_streams) e->aux_stream = e->pre_stream = e->ana_stream = e->fin_stream = e->eq_stream = nullptr;
  if (e->lim_stream) (void)hipStreamDestroy(e->lim_stream);
  if (e->fin_stream) (void)hipStreamDestroy(e->fin_stream);
  if (e->rnn_stream) (void)hipStreamDestroy(e->rnn_stream);
  if (e->eq_stream) (void)hipStreamDestroy(e->eq_stream);
  if (e->syn_stream) (void)hipStreamDestroy(e->syn_stream);
  if (e->aux_stream) (void)hipStreamDestroy(e->aux_stream);
  if (e->pre_stream) (void)hipStreamDestroy(e->pre_stream);
  if (e->ana_stream) (void)hipStreamDestroy(e->ana_stream);
  if (e->supp.d_blob || e->supp.d_state || e->supp.d_xh) {
    (void)hipSetDevice(e->device);
    e->supp.release_all();
  }
  delete e;
}

int af_engine_reset(af_engine *e) {
  if (!e) return fail(AF_ERR_INVALID_ARGUMENT, "engine is null");
  if (e->started) {
    AF_HIP(hipSetDevice(e->device));
    AF_HIP(hipDeviceSynchronize());
    (void)collect_retired(e, true);
  }
  e->started = false;
  e->uploaded_valid = e->uploaded_pre_valid = e->uploaded_de_valid = false;
  e->pipe.decided = false;
  e->params_dirty = true;
  e->samples_processed = 0;
  e->last_blocks = 0;
  e->pending = 0;
  e->last_output_samples = 0;
  e->trace_frames = 0;
  return AF_OK;
}

int32_t af_engine_n_streams(const af_engine *e) { return e ? e->n_streams : 0; }

#define AF_SETTER(expr)              \
  do {                               \
    int rc__ = require_config(e);    \
    if (rc__) return rc__;           \
    expr;                            \
    return AF_OK;                    \
  } while (0)

int af_engine_set_deesser_enabled(af_engine *e, int32_t on) { AF_SETTER(cur(e).deesser_enabled = cur(e).deesser.enabled = on != 0); }
int af_engine_set_eq_enabled(af_engine *e, int32_t on) { AF_SETTER(cur(e).eq_enabled = cur(e).eq.enabled = on != 0); }
int af_engine_set_compressor_enabled(af_engine *e, int32_t on) { AF_SETTER(cur(e).compressor_enabled = cur(e).compressor.enabled = on != 0); }
int af_engine_set_limiter_enabled(af_engine *e, int32_t on) { AF_SETTER(cur(e).limiter_enabled = cur(e).limiter.enabled = on != 0); }
int af_engine_set_eq_before_deesser(af_engine *e, int32_t on) { AF_SETTER(cur(e).eq_before_deesser = on != 0); }
// the front end belongs to the engine, not to a preset (with the suppressor on it runs in the suppressor's pre-pass)
#define AF_ALL_PRESETS(stmt)                                     \
  do {                                                           \
    { af::ChainProto &p = e->proto; stmt; }                      \
    for (af_preset &ps__ : e->extra_presets) { af::ChainProto &p = ps__.proto; stmt; } \
  } while (0)
int af_engine_set_input_scrub_enabled(af_engine *e, int32_t on) { AF_SETTER(AF_ALL_PRESETS(p.input_scrub = on != 0)); }
int af_engine_set_input_clamp_enabled(af_engine *e, int32_t on) { AF_SETTER(AF_ALL_PRESETS(p.input_clamp = on != 0)); }
int af_engine_set_prefilter_enabled(af_engine *e, int32_t on, int32_t hp) {
  AF_SETTER(AF_ALL_PRESETS((p.dc_block = on != 0, p.pre_highpass = on != 0 && hp != 0)));
}
int af_engine_set_control_block_samples(af_engine *e, int32_t n) {
  if (n < 1 || n > 8192) return fail(AF_ERR_INVALID_ARGUMENT, "control block must be in [1, 8192] samples");
  AF_SETTER(AF_ALL_PRESETS(p.control_block = n));
}

int af_eq_set_band_frequency(af_engine *e, int32_t band, double hz) {
  if (int rc = check_band(band)) return rc;
  AF_SETTER(cur(e).eq.set_band_frequency(band, hz));
}
int af_eq_set_band_gain(af_engine *e, int32_t band, double db) {
  if (int rc = check_band(band)) return rc;
  AF_SETTER(cur(e).eq.set_band_gain(band, db));
}
int af_eq_set_band_q(af_engine *e, int32_t band, double q) {
  if (int rc = check_band(band)) return rc;
  AF_SETTER(cur(e).eq.set_band_q(band, q));
}
int af_eq_set_band_config(af_engine *e, int32_t band, const af_eq_band_config *c) {
  if (int rc = check_band(band)) return rc;
  if (!c) return fail(AF_ERR_INVALID_ARGUMENT, "config is null");
  AF_SETTER(cur(e).eq.set_band_config(band, to_cfg(*c)));
}
int af_eq_reset(af_engine *e) { AF_SETTER(cur(e).eq.reset()); }
int af_eq_band_config_validate(const af_eq_band_config *c, int32_t index, double sample_rate) {
  if (!c) return fail(AF_ERR_INVALID_ARGUMENT, "config is null");
  if (c->filter_type < 0 || c->filter_type > 5)
    return fail(AF_ERR_INVALID_ARGUMENT, "band %d has unsupported EQ filter type id: %d", index, c->filter_type);
  const std::string msg = af::eq_validate(to_cfg(*c), index, sample_rate);
  if (!msg.empty()) return fail(AF_ERR_INVALID_ARGUMENT, "%s", msg.c_str());
  return AF_OK;
}

int af_compressor_set_threshold(af_engine *e, double v) { AF_SETTER(cur(e).compressor.set_threshold(v)); }
int af_compressor_set_ratio(af_engine *e, double v) { AF_SETTER(cur(e).compressor.set_ratio(v)); }
int af_compressor_set_attack_time(af_engine *e, double v) { AF_SETTER(cur(e).compressor.set_attack_time(v)); }
int af_compressor_set_release_time(af_engine *e, double v) { AF_SETTER(cur(e).compressor.set_release_time(v)); }
int af_compressor_set_makeup_gain(af_engine *e, double v) { AF_SETTER(cur(e).compressor.set_makeup_gain(v)); }
int af_compressor_set_adaptive_release(af_engine *e, int32_t on) { AF_SETTER(cur(e).compressor.set_adaptive_release(on != 0)); }
int af_compressor_set_base_release_time(af_engine *e, double v) { AF_SETTER(cur(e).compressor.set_base_release_time(v)); }
int af_compressor_set_auto_makeup_enabled(af_engine *e, int32_t on) { AF_SETTER(cur(e).compressor.set_auto_makeup_enabled(on != 0)); }
int af_compressor_set_target_lufs(af_engine *e, double v) { AF_SETTER(cur(e).compressor.set_target_lufs(v)); }
int af_compressor_set_sidechain_highpass_enabled(af_engine *e, int32_t on) { AF_SETTER(cur(e).compressor.set_sidechain_highpass_enabled(on != 0)); }

int af_compressor_set_noise_reference_reliability(af_engine *e, double v) { AF_SETTER(cur(e).compressor.set_noise_reference_reliability(v)); }

int af_compressor_set_activity_evidence(af_engine *e, const double *vad_probabilities, int64_t n_blocks, int32_t per_stream,
                                        double vad_reliability, double noise_floor_db, double live_noise_reliability) {
  if (!e) return fail(AF_ERR_INVALID_ARGUMENT, "engine is null");
  if (n_blocks < 0 || (n_blocks > 0 && !vad_probabilities)) return fail(AF_ERR_INVALID_ARGUMENT, "bad VAD array");
  auto unit = [](double v) { return std::isfinite(v) ? af::clampd(v, 0.0, 1.0) : 0.0; };  // compressor.rs:516-518
  e->has_evidence = n_blocks > 0;
  e->vad_reliability = unit(vad_reliability);
  e->noise_floor_db = std::isfinite(noise_floor_db) ? noise_floor_db : 1.0;  // out of [-120, 0] disables it
  e->live_noise_reliability = unit(live_noise_reliability);
  e->vad_blocks = n_blocks;
  e->params_dirty = true;
  if (e->started) {  // ChainParams carries the three scalars
    e->host_params.comp.vad_reliability = e->vad_reliability;
    e->host_params.comp.noise_floor_db = e->noise_floor_db;
    e->host_params.comp.live_noise_reliability = e->live_noise_reliability;
    e->host_params.comp.has_evidence = e->has_evidence ? 1 : 0;
  }
  if (n_blocks == 0) return AF_OK;
  AF_HIP(hipSetDevice(e->device));
  const int64_t total = n_blocks * e->n_streams;
  if (total > e->vad_capacity) {
    if (e->d_vad) AF_HIP(hipFree(e->d_vad));
    AF_HIP(hipMalloc(&e->d_vad, sizeof(double) * total));
    e->vad_capacity = total;
  }
  if (per_stream) {
    AF_HIP(hipMemcpy(e->d_vad, vad_probabilities, sizeof(double) * total, hipMemcpyHostToDevice));
  } else {
    std::vector<double> expanded((size_t)total);
    for (int64_t b = 0; b < n_blocks; ++b) std::fill_n(expanded.begin() + b * e->n_streams, e->n_streams, vad_probabilities[b]);
    AF_HIP(hipMemcpy(e->d_vad, expanded.data(), sizeof(double) * total, hipMemcpyHostToDevice));
  }
  return AF_OK;
}

int af_limiter_set_ceiling(af_engine *e, double v) { AF_SETTER(cur(e).limiter.set_ceiling(v)); }
int af_limiter_set_release_time(af_engine *e, double v) { AF_SETTER(cur(e).limiter.set_release_time(v)); }
int af_limiter_set_lookahead_ms(af_engine *e, double v) { AF_SETTER(cur(e).limiter.set_lookahead_ms(v)); }
double af_limiter_ceiling_db(const af_engine *e) { return e ? cur(e).limiter.ceiling_db : 0.0; }
int32_t af_limiter_lookahead_samples(const af_engine *e) { return e ? cur(e).limiter.lookahead_samples : 0; }

int af_true_peak_limiter_set_release_ms(af_engine *e, float ms) { AF_SETTER(cur(e).tp_limiter.set_release_ms(ms)); }

int af_deesser_set_auto_enabled(af_engine *e, int32_t on) { AF_SETTER(cur(e).deesser.auto_enabled = on != 0); }
int af_deesser_set_auto_amount(af_engine *e, double v) { AF_SETTER(cur(e).deesser.set_auto_amount(v)); }
int af_deesser_set_low_cut_hz(af_engine *e, double v) { AF_SETTER(cur(e).deesser.set_low_cut_hz(v)); }
int af_deesser_set_high_cut_hz(af_engine *e, double v) { AF_SETTER(cur(e).deesser.set_high_cut_hz(v)); }
int af_deesser_set_threshold_db(af_engine *e, double v) { AF_SETTER(cur(e).deesser.set_threshold_db(v)); }
int af_deesser_set_ratio(af_engine *e, double v) { AF_SETTER(cur(e).deesser.set_ratio(v)); }
int af_deesser_set_attack_ms(af_engine *e, double v) { AF_SETTER(cur(e).deesser.set_attack_ms(v)); }
int af_deesser_set_release_ms(af_engine *e, double v) { AF_SETTER(cur(e).deesser.set_release_ms(v)); }
int af_deesser_set_max_reduction_db(af_engine *e, double v) { AF_SETTER(cur(e).deesser.set_max_reduction_db(v)); }

// ---- RNNoise suppressor (rust-core/src/dsp/rnnoise.rs) ----
int af_engine_set_suppressor_enabled(af_engine *e, int32_t on) { AF_SETTER(e->supp.enabled = on != 0); }
int af_engine_set_suppressor_strength(af_engine *e, float strength) {  // rnnoise.rs:67-72; allowed while streaming
  if (!e) return fail(AF_ERR_INVALID_ARGUMENT, "engine is null");
  e->supp.strength = af::clampf(strength, 0.0f, 1.0f);
  return AF_OK;
}
int af_suppressor_set_raw_protocol(af_engine *e, int32_t on) { AF_SETTER(e->supp.raw_protocol = on != 0); }
int af_suppressor_set_synthetic_weights(af_engine *e, uint64_t seed) {
  AF_SETTER((af::synthetic_weights(e->supp.weights, seed), e->supp.weights_dirty = true));
}
int af_suppressor_load_weights(af_engine *e, const int8_t *blob, size_t bytes) {
  if (!blob || bytes != sizeof(af::RnnWeightsI8))
    return fail(AF_ERR_INVALID_ARGUMENT, "weight blob must be %zu bytes (the fifteen int8 arrays of the RNNoise model)",
                sizeof(af::RnnWeightsI8));
  AF_SETTER((std::memcpy(&e->supp.weights, blob, bytes), e->supp.weights_dirty = true));
}
int32_t af_suppressor_latency_samples(const af_engine *) { return af::kRnnFrame; }  // rnnoise.rs:313-315
// test tap: one (frame, stream) record of the LAST window: Ex Ep Exp feat[44] gains_raw gains silence pitch, then X, P
int af_suppressor_debug_read(af_engine *e, int32_t frame, int32_t stream, float *rec_out, float *x_out, float *p_out) {
  if (!e || !e->supp.d_rec) return fail(AF_ERR_STATE, "no suppressor window has run");
  if (frame < 0 || frame >= e->supp.ws_frames || stream < 0 || stream >= e->n_streams)
    return fail(AF_ERR_INVALID_ARGUMENT, "frame/stream out of range");
  AF_HIP(hipSetDevice(e->device));
  AF_HIP(hipDeviceSynchronize());
  const size_t cell = (size_t)frame * e->n_streams + stream;
  AF_HIP(hipMemcpy(rec_out, e->supp.d_rec + cell, sizeof(af::SuppFrameRec), hipMemcpyDeviceToHost));
  if (x_out) AF_HIP(hipMemcpy(x_out, e->supp.d_X + cell * af::kRnnFreq, sizeof(float2) * af::kRnnFreq, hipMemcpyDeviceToHost));
  if (p_out) AF_HIP(hipMemcpy(p_out, e->supp.d_P + cell * af::kRnnFreq, sizeof(float2) * af::kRnnFreq, hipMemcpyDeviceToHost));
  return AF_OK;
}

// ---- presets: several chain configurations in one engine, one per 64-stream group
int af_engine_set_preset_count(af_engine *e, int32_t n) {
  if (n < 1 || n > 256) return fail(AF_ERR_INVALID_ARGUMENT, "preset count must be in [1, 256]");
  if (int rc = require_config(e)) return rc;
  while ((int)e->extra_presets.size() > n - 1) e->extra_presets.pop_back();
  while ((int)e->extra_presets.size() < n - 1) {  // a fresh OfflineDspBlockProcessor::new(sample_rate), block_processor.rs:46-60
    e->extra_presets.emplace_back(e->proto.sample_rate);
    af::ChainProto &np = e->extra_presets.back().proto;
    np.control_block = e->proto.control_block;
    np.input_scrub = e->proto.input_scrub;
    np.input_clamp = e->proto.input_clamp;
    np.dc_block = e->proto.dc_block;
    np.pre_highpass = e->proto.pre_highpass;
  }
  if (e->current_preset >= n) e->current_preset = 0;
  for (int32_t &g : e->group_preset)
    if (g >= n) g = 0;
  return AF_OK;
}
int32_t af_engine_preset_count(const af_engine *e) { return e ? 1 + (int32_t)e->extra_presets.size() : 0; }
int af_engine_select_preset(af_engine *e, int32_t preset) {
  if (!e) return fail(AF_ERR_INVALID_ARGUMENT, "engine is null");
  if (preset < 0 || preset > (int32_t)e->extra_presets.size()) return fail(AF_ERR_INVALID_ARGUMENT, "preset %d does not exist", preset);
  e->current_preset = preset;
  return AF_OK;
}
int af_engine_assign_presets(af_engine *e, const int32_t *preset_of_group, int32_t n_groups) {
  if (!e || !preset_of_group) return fail(AF_ERR_INVALID_ARGUMENT, "null argument");
  if (n_groups != (e->n_streams + 63) / 64)
    return fail(AF_ERR_INVALID_ARGUMENT, "expected one preset index per group of 64 streams (%d), got %d", (e->n_streams + 63) / 64, n_groups);
  for (int32_t g = 0; g < n_groups; ++g)
    if (preset_of_group[g] < 0 || preset_of_group[g] > (int32_t)e->extra_presets.size())
      return fail(AF_ERR_INVALID_ARGUMENT, "group %d: preset %d does not exist", g, preset_of_group[g]);
  if (int rc = require_config(e)) return rc;
  e->group_preset.assign(preset_of_group, preset_of_group + n_groups);
  return AF_OK;
}

int af_engine_set_kernel(af_engine *e, int32_t kernel) {
  if (!e) return fail(AF_ERR_INVALID_ARGUMENT, "engine is null");
  if (kernel < AF_KERNEL_AUTO || kernel > AF_KERNEL_ROLES) return fail(AF_ERR_INVALID_ARGUMENT, "unknown kernel id %d", kernel);
  e->kernel = kernel;
  return AF_OK;
}
int af_engine_set_ring_variant(af_engine *e, int32_t waves, int32_t chunk) {
  if (!e) return fail(AF_ERR_INVALID_ARGUMENT, "engine is null");
  const int v = waves * 100 + chunk;
  if (v != 0 && v != 1604 && v != 1602 && v != 804 && v != 802 && v != 1204 && v != 1202)
    return fail(AF_ERR_INVALID_ARGUMENT, "no token-ring kernel is built for %d waves x %d-sample chunks", waves, chunk);
  e->ring_variant = v;
  return AF_OK;
}
int af_engine_last_kernel(const af_engine *e) { return e ? e->last_kernel_used : 0; }
int af_engine_set_timing_enabled(af_engine *e, int32_t on) {
  if (!e) return fail(AF_ERR_INVALID_ARGUMENT, "engine is null");
  e->timing = on != 0;
  return AF_OK;
}

int af_engine_process_device(af_engine *e, const float *in, float *out, int64_t n_samples, int64_t stream_stride,
                             int32_t layout, void *hip_stream) {
  if (!e) return fail(AF_ERR_INVALID_ARGUMENT, "engine is null");
  if (n_samples < 0) return fail(AF_ERR_INVALID_ARGUMENT, "n_samples must be >= 0");
  if (layout != AF_LAYOUT_STREAM_MAJOR && layout != AF_LAYOUT_TIME_MAJOR)
    return fail(AF_ERR_INVALID_ARGUMENT, "unknown layout %d", layout);
  if (n_samples > 0 && (!in || !out)) return fail(AF_ERR_INVALID_ARGUMENT, "audio pointers are null");
  const int64_t min_stride = layout == AF_LAYOUT_STREAM_MAJOR ? n_samples : e->n_streams;
  if (stream_stride < min_stride) return fail(AF_ERR_INVALID_ARGUMENT, "stream_stride %lld is smaller than %lld",
                                              (long long)stream_stride, (long long)min_stride);
  if (int rc = ensure_started(e)) return rc;
  hipStream_t stream = (hipStream_t)hip_stream;
  // ---- Everything that can refuse the call is checked before the engine's frame ring, its buffers or a stream are touched: a
  // refused call leaves af_engine_pending_input and the audio state as they were.
  // RNNoise frame buffering (rnnoise.rs:114-164: push_samples -> process_frames -> pop): the suppressor eats whole 480-sample
  // frames; what a call leaves over waits in the engine for the next call, and a call returns the whole frames that are
  // complete by then: floor((pending + n) / 480) * 480 samples per stream, which may be 0 or exceed n.
  const float *src = in;
  int64_t src_stride = stream_stride;
  const int64_t n_in = n_samples;
  int64_t n_run = n_samples, rem = 0;
  if (e->supp.enabled) {
    if (layout != AF_LAYOUT_STREAM_MAJOR) return fail(AF_ERR_UNSUPPORTED, "the suppressor needs stream-major audio");
    const int64_t total = e->pending + n_in;
    n_run = (total / af::kRnnFrame) * af::kRnnFrame;
    rem = total - n_run;
    if (n_run > stream_stride)
      return fail(AF_ERR_INVALID_ARGUMENT, "this call completes %lld samples per stream (%d were pending): stream_stride %lld is too small",
                  (long long)n_run, e->pending, (long long)stream_stride);
  }
  const int cb = e->host_params.control_block;
  const int64_t blocks = (n_run + cb - 1) / cb;
  if (n_run > 0 && (e->host_params.flags & af::kFlagCompressor) && e->host_params.comp.auto_makeup_enabled && e->has_evidence &&
      e->vad_blocks != blocks)
    return fail(AF_ERR_INVALID_ARGUMENT, "expected %lld VAD probabilities at the control cadence, got %lld",
                (long long)blocks, (long long)e->vad_blocks);
  if (n_run > 0 && !e->pipe.decided) {  // first call after a reset: which form of the chain this engine runs
    af::ChainParams probe = e->host_params;
    if (e->supp.enabled) probe.flags &= ~(af::kFlagInputClamp | af::kFlagDcBlock | af::kFlagPreHighpass | af::kFlagInputScrub);
    const bool serves = stage_pipe_serves(e, probe, layout);
    static const int env_staged = [] {  // AF_STAGED=0 / 1: keep AUTO off / on the stage pipeline (A/B runs)
      const char *env = std::getenv("AF_STAGED");
      return env ? std::atoi(env) : -1;
    }();
    if (e->kernel == AF_KERNEL_STAGED && !serves)
      return fail(AF_ERR_UNSUPPORTED, "the stage pipeline does not build this configuration (EQ-before-de-esser order, front end without the "
                                      "suppressor, more than 16 EQ sections, presets that differ in which stages run, time-major audio)");
    // (with the de-esser at any batch: its lane-per-stream form takes 417 ms per 2 s of audio whatever the batch)
    // (and several presets with auto-makeup: the token-ring kernel's pre-pass pair is a single-preset build)
    const bool deesser_staged = (probe.flags & af::kFlagDeesser) != 0 ||
                                (!e->extra_presets.empty() && (probe.flags & af::kFlagCompressor) && probe.comp.auto_makeup_enabled);
    e->pipe.active = serves && (e->kernel == AF_KERNEL_STAGED || (e->kernel == AF_KERNEL_AUTO && env_staged != 0 &&
                                 (deesser_staged || e->n_streams <= (e->supp.enabled ? kStagedAutoMaxStreamsBehindSuppressor : kStagedAutoMaxStreams))) ||
                                (e->kernel == AF_KERNEL_AUTO && env_staged > 0));
    e->pipe.decided = true;
    if (e->pipe.active)
      if (int rc = stage_pipe_clear(e)) return rc;
  }
  // ---- accepted: from here on the call only fails on a backend error
  e->last_stream = stream;
  (void)collect_retired(e, false);
  if (e->supp.enabled) {
    const int64_t B = e->n_streams;
    if (e->pending > 0 || rem > 0) {
      if (!e->d_pending) AF_HIP(hipMalloc(&e->d_pending, sizeof(float) * af::kRnnFrame * B));
      const size_t f4 = sizeof(float);
      if (n_run > 0) {
        int64_t cap = e->asm_capacity * (int64_t)f4;  // (scratch of one call: grown geometrically, the old buffer retired)
        if (int rc = grow_device(e, reinterpret_cast<void **>(&e->d_asm), &cap, B * n_run * (int64_t)f4, stream)) return rc;
        e->asm_capacity = cap / (int64_t)f4;
        // [pending | head of this call] -> whole frames; the tail of this call waits (copied before anything writes `out`,
        // which may alias `in`)
        if (e->pending > 0)
          AF_HIP(hipMemcpy2DAsync(e->d_asm, f4 * n_run, e->d_pending, f4 * af::kRnnFrame, f4 * e->pending, B, hipMemcpyDeviceToDevice, stream));
        AF_HIP(hipMemcpy2DAsync(e->d_asm + e->pending, f4 * n_run, in, f4 * stream_stride, f4 * (n_run - e->pending), B,
                                hipMemcpyDeviceToDevice, stream));
        if (rem > 0)
          AF_HIP(hipMemcpy2DAsync(e->d_pending, f4 * af::kRnnFrame, in + (n_in - rem), f4 * stream_stride, f4 * rem, B,
                                  hipMemcpyDeviceToDevice, stream));
        src = e->d_asm;
        src_stride = n_run;
      } else if (n_in > 0) {
        AF_HIP(hipMemcpy2DAsync(e->d_pending + e->pending, f4 * af::kRnnFrame, in, f4 * stream_stride, f4 * n_in, B,
                                hipMemcpyDeviceToDevice, stream));
      }
      e->pending = (int)rem;
    }
    n_samples = n_run;
  }
  e->last_output_samples = n_samples;
  e->trace_frames = 0;
  e->ev_cursor = 0;
  e->last_blocks = blocks;
  e->last_kernel_ms = 0.0;
  e->last_launches = 0;
  if (n_samples == 0) return AF_OK;
  const int64_t rows = blocks * e->n_streams;
  {
    int64_t cap = e->stats_capacity * (int64_t)sizeof(af::BlockStats);
    if (int rc = grow_device(e, reinterpret_cast<void **>(&e->d_stats), &cap, rows * (int64_t)sizeof(af::BlockStats), stream)) return rc;
    e->stats_capacity = cap / (int64_t)sizeof(af::BlockStats);
  }
  if (e->timing) {
    if (!e->ev_start) {
      AF_HIP(hipEventCreate(&e->ev_start));
      AF_HIP(hipEventCreate(&e->ev_stop));
      AF_HIP(hipEventCreate(&e->ev_mid));
    }
    AF_HIP(hipEventRecord(e->ev_start, stream));
  }
  for (auto &pr : e->chain_ms_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  e->chain_ms_events.clear();

  if (!e->supp.enabled && e->pipe.active) {
    // ---- the chain as a pipeline of stage kernels over windows of whole control blocks (af_stages.hip)
    // (a launch step costs ~20 us, the pipeline's fill is depth x window time: 960 samples 45.7 ms per 10 s at 256 streams,
    // 1920 41.2, 2880 39.7, 4800 42.7, 9600 40.3)
    int64_t tw = (int64_t)cb * std::max<int64_t>(1, 2880 / cb);
    if (const char *env = std::getenv("AF_STAGE_WINDOW")) tw = (int64_t)cb * std::max<int64_t>(1, std::atoll(env) / cb);
    if (int rc = stage_pipe_prepare(e, std::max<int64_t>(tw, e->pipe.tw_max))) return rc;
    e->pipe.strip = 0;
    if (int rc = stage_chain_params(e, stream)) return rc;  // what the stage kernels read (everything but the EQ sections)
    e->last_kernel_used = AF_KERNEL_STAGED;
    {
      // one launch step per window on the caller's stream: step j runs every stage on the window it has reached
      const StagePlan plan = stage_plan(e->host_params);
      std::vector<af::DiagWin> wins;
      int64_t blocks_at = 0;
      for (int64_t t0 = 0; t0 < n_samples; t0 += tw) {
        const int64_t n_w = std::min<int64_t>(tw, n_samples - t0);
        af::DiagWin wd{};
        wd.n0 = e->samples_processed + t0;
        wd.n = n_w;
        wd.stats = e->d_stats + blocks_at * e->n_streams;
        wd.mk = e->pipe.d_mk + ((e->pipe.windows + (int64_t)wins.size()) % af_engine::StagePipe::kMkSets) * e->pipe.mk_rows;
        wd.bp = e->pipe.d_bp + ((e->pipe.windows + (int64_t)wins.size()) % af_engine::StagePipe::kBpSets) * e->pipe.mk_rows;
        wd.vad = e->has_evidence ? e->d_vad + blocks_at * e->n_streams : nullptr;
        wd.in = in + t0;
        wd.out = out + t0;
        wins.push_back(wd);
        blocks_at += (n_w + cb - 1) / cb;
      }
      e->pipe.call_stride = stream_stride;
      AF_HIP(hipMemsetAsync(e->d_stats, 0, sizeof(af::BlockStats) * rows, stream));
      const int64_t steps = (int64_t)wins.size() + plan.depth;
      for (int64_t j = 0; j < steps; ++j) {
        if (j < (int64_t)wins.size()) {  // window j enters: its EQ stage reads the section parameters as they stand now
          bool crossfade = false;
          if (int rc = stage_diag_eq_params(e, stream, &crossfade, &wins[(size_t)j].eq_slot)) return rc;
          wins[(size_t)j].eq_crossfade = crossfade ? 1 : 0;
          advance_crossfades(e, wins[(size_t)j].n);
        }
        if (int rc = stage_diag_step(e, e->host_params, plan, wins, j, stream)) return rc;
      }
      e->pipe.windows += (int64_t)wins.size();
      if (e->timing) {
        AF_HIP(hipEventRecord(e->ev_mid, stream));
        AF_HIP(hipEventRecord(e->ev_stop, stream));
      }
      e->samples_processed += n_samples;
      return AF_OK;
    }
  }
  if (!e->supp.enabled) {
    // ---- Large batches without the suppressor (round 3): the chain can use one CU per 64 streams and nothing else, so the EQ
    // -- a quarter of the token-ring kernel's time -- runs as the systolic kernel on the CUs the chain leaves idle, window by
    // window, and the chain is ONE launch that follows it through the ready counter (the form the suppressor's pipeline
    // uses).  Taken when the streams can be CU-partitioned and the EQ kernel serves the configuration.
    {
      const af::ChainParams &hp = e->host_params;
      static const bool eq_offload_on = [] {
        const char *env = std::getenv("AF_EQ_OFFLOAD");
        return !env || std::atoi(env) != 0;
      }();
      const bool auto_mk = (hp.flags & af::kFlagCompressor) && hp.comp.auto_makeup_enabled;
      const int64_t window = (int64_t)cb * std::max<int64_t>(1, 9600 / cb);
      bool offload = one_launch_calls_enabled() && eq_offload_on && (e->kernel == AF_KERNEL_AUTO || e->kernel == AF_KERNEL_PHASED) &&
                     (e->ring_variant == 0 || e->ring_variant == 1604) && e->extra_presets.empty() && roles_mode() == 0 &&
                     layout == AF_LAYOUT_STREAM_MAJOR && (hp.flags & af::kFlagEq) && hp.n_eq_sections > 0 && hp.n_eq_sections <= 16 &&
                     !(hp.flags & (af::kFlagDeesser | af::kFlagDcBlock | af::kFlagPreHighpass | af::kFlagPrePass)) &&
                     af::ring_kernel_dynamic_lds(hp.n_eq_sections, hp.lim.lookahead_samples, true) <= af::kMaxLdsBytes &&
                     n_samples >= 2 * window && !std::getenv("AF_SERIAL_STREAMS");
      if (offload) {
        if (int rc = ensure_side_streams(e, stream)) return rc;
        offload = e->partition_chain_cus > 0 && !e->borrowed_streams;
      }
      if (offload) {
        if (auto_mk) {
          int64_t cap = e->block_power_capacity * (int64_t)sizeof(double);
          if (int rc = grow_device(e, reinterpret_cast<void **>(&e->d_block_power), &cap, rows * (int64_t)sizeof(double), stream)) return rc;
          e->block_power_capacity = cap / (int64_t)sizeof(double);
        }
        AF_HIP(hipMemsetAsync(e->d_stats, 0, sizeof(af::BlockStats) * rows, stream));
        if (!e->d_ready) AF_HIP(hipMalloc(&e->d_ready, sizeof(int64_t)));
        AF_HIP(hipMemsetAsync(e->d_ready, 0, sizeof(int64_t), stream));
        hipEvent_t ev;
        if (int rc = engine_event(e, &ev)) return rc;
        AF_HIP(hipEventRecord(ev, stream));
        AF_HIP(hipStreamWaitEvent(e->aux_stream, ev, 0));
        AF_HIP(hipStreamWaitEvent(e->eq_stream, ev, 0));
        af::ChainParams run_p = hp;  // the EQ kernel scrubs / clamps the input and keeps the block input statistics
        run_p.flags = (run_p.flags & ~(af::kFlagEq | af::kFlagInputScrub | af::kFlagInputClamp)) | af::kFlagInputDone;
        if (int rc = launch_chain_segment(e, run_p, true, out, out, n_samples, stream_stride, layout, e->samples_processed, e->d_stats,
                                          e->has_evidence ? e->d_vad : nullptr, e->aux_stream, stream, /*stats_cleared=*/true,
                                          auto_mk ? e->d_block_power : nullptr, e->d_ready))
          return rc;
        if (!e->d_params_eq || e->eq_params_presets != 1) {
          if (e->d_params_eq) AF_HIP(hipFree(e->d_params_eq));
          e->d_params_eq = nullptr;
          AF_HIP(hipMalloc(&e->d_params_eq, sizeof(af::ChainParams) * kEqParamSlots));
          e->eq_params_presets = 1;
          e->uploaded_eq.clear();
        }
        int64_t blocks_done = 0;
        for (int64_t seg0 = 0; seg0 < n_samples; seg0 += window) {
          const int64_t seg_n = std::min<int64_t>(window, n_samples - seg0);
          std::vector<af::ChainParams> run_eq(1, e->host_params);  // (as the crossfade counters stand at this window)
          bool xf_w = false;
          for (int j = 0; j < run_eq[0].n_eq_sections; ++j) xf_w = xf_w || run_eq[0].eq[j].xf_remaining > 0;
          if (e->uploaded_eq.size() != 1 || std::memcmp(e->uploaded_eq.data(), run_eq.data(), sizeof(af::ChainParams)) != 0) {
            e->uploaded_eq = run_eq;
            if (int rc = stage_upload(e, e->d_params_eq, run_eq.data(), 1, e->eq_stream)) return rc;
          }
          AF_HIP(af::launch_eq_systolic(e->d_params_eq, nullptr, e->d_st64, in + seg0, out + seg0, nullptr, nullptr, 0, 0,
                                        e->d_stats + blocks_done * e->n_streams, xf_w, seg_n, stream_stride, e->n_streams, e->eq_stream,
                                        auto_mk ? e->d_block_power + blocks_done * e->n_streams : nullptr));  // (the systolic form: here the EQ's own latency per window is what the chain follows)
          AF_HIP(af::launch_chain_publish_ready(e->d_ready, seg0 + seg_n, e->eq_stream));
          e->last_launches += 2;
          advance_crossfades(e, seg_n);
          blocks_done += (seg_n + cb - 1) / cb;
        }
        for (hipStream_t side : {e->aux_stream, e->eq_stream}) {
          if (int rc = engine_event(e, &ev)) return rc;
          AF_HIP(hipEventRecord(ev, side));
          AF_HIP(hipStreamWaitEvent(stream, ev, 0));
        }
        if (e->timing) {
          AF_HIP(hipEventRecord(e->ev_mid, stream));
          AF_HIP(hipEventRecord(e->ev_stop, stream));
        }
        e->samples_processed += n_samples;
        return AF_OK;
      }
    }
    int rc = launch_chain_segment(e, e->host_params, false, in, out, n_samples, stream_stride, layout, e->samples_processed,
                                  e->d_stats, e->has_evidence ? e->d_vad : nullptr, stream, stream);
    if (rc) return rc;
    if (e->timing) {
      AF_HIP(hipEventRecord(e->ev_mid, stream));  // no suppressor: everything is chain time
      AF_HIP(hipEventRecord(e->ev_stop, stream));
    }
    e->samples_processed += n_samples;
    return AF_OK;
  }

  // ---- RNNoise suppressor ahead of the chain (realtime order, dsp_loop.rs:1222-1250,1521-1599).
  // The call is cut into windows of frames and runs as a four-stage pipeline over them, one HIP stream each:
  //   pre stream    : window w+2's sample-serial pre-pass (front end + model-input high-pass; 64 waves whose
  //                   duration is set by recurrence latency, so it costs the chip almost nothing)
  //   analysis      : window w+1's spectra and pitch search (the pitch kernel walks each stream's frames in
  //                   order, one wave per stream: latency bound, it leaves most issue slots free)
  //   caller stream : window w's pitch-aligned spectra, network, resynthesis, overlap-add
  //   chain stream  : window w-1's chain launch (64 streams per workgroup, a quarter of the CUs at batch 4096)
  // ordered by events; buffers that cross a stage boundary rotate (af_suppressor_host.hpp).
  if (e->supp.weights_dirty) AF_HIP(e->supp.upload());
  af::ChainParams run = e->host_params;
  bool run_modified = false;
  const uint32_t front = af::kFlagInputClamp | af::kFlagDcBlock | af::kFlagPreHighpass;
  const uint32_t front_flags = run.flags & front;
  if (front_flags) {
    // the realtime front end (clamp + DC block + 80 Hz HP, routing.rs:802-843) runs inside the suppressor's
    // own sample-serial pre-pass, so the chain launches must not repeat it
    run.flags &= ~(front | af::kFlagInputScrub);
    run_modified = true;
  }
  const int64_t frames = n_samples / af::kRnnFrame;
  // a window must hold whole control blocks, or block boundaries (hence per-block semantics) would move
  int64_t unit = 1;
  while ((unit * af::kRnnFrame) % cb != 0) ++unit;
  int window_frames = e->supp_window_frames;
  if (const char *env = std::getenv("AF_SUPP_WINDOW_FRAMES")) window_frames = std::max(1, std::atoi(env));  // tuning runs
  // Only the last control block of a call can be short, so the call is scheduled as aligned windows over its whole control
  // blocks plus one short final window for a ragged end: block boundaries do not move and the pipeline keeps its overlap.
  const int64_t aligned = (frames / unit) * unit, ragged = frames - aligned;
  int64_t window = std::max<int64_t>(unit, (window_frames / unit) * unit);
  window = std::min<int64_t>(window, std::max<int64_t>(aligned, unit));
  // Window schedule.  The first chain launch cannot start before one window has been through the pre-pass, the analysis
  // and the synthesis, and the last chain launch runs after everything else is done: with uniform windows that is ~2.5
  // window times of a 20-window call during which most of the chip idles.  So the call opens with short windows that
  // double up to the full size and closes with the mirror image (every size a whole number of control blocks).
  std::vector<int64_t> win_f0, win_nf;
  {
    static const bool ramp = [] {
      const char *env = std::getenv("AF_SUPP_RAMP");
      return !env || std::atoi(env) != 0;
    }();
    std::vector<int64_t> up;
    static const std::vector<int64_t> up_env = [] {  // AF_SUPP_RAMP_LIST=4,4,8,8,16: the opening windows, in frames (tuning runs)
      std::vector<int64_t> v;
      if (const char *env = std::getenv("AF_SUPP_RAMP_LIST"))
        for (const char *p = env; *p;) {
          char *end = nullptr;
          const long n = std::strtol(p, &end, 10);
          if (end == p) break;
          if (n > 0) v.push_back(n);
          p = *end ? end + 1 : end;
        }
      return v;
    }();
    // The mirror image at the end of the call shortens what runs after the suppressor's last kernel -- when that is the stage
    // pipeline emptying.  Behind the token-ring kernel the chain is the longer side and trails the suppressor by more than a
    // window anyway: there the small windows only cost launches (189.3 against 190.0 ms per bench step).  AF_SUPP_RAMP_END=0 / 1.
    static const int ramp_down_env = [] {
      const char *env = std::getenv("AF_SUPP_RAMP_END");
      return env ? std::atoi(env) : -1;
    }();
    const bool ramp_down = ramp_down_env >= 0 ? ramp_down_env != 0 : e->pipe.active;
    if (!up_env.empty()) {
      for (int64_t n : up_env) up.push_back(std::min<int64_t>(window, ((n + unit - 1) / unit) * unit));
    } else {
      for (int64_t n = ((4 + unit - 1) / unit) * unit; n < window; n *= 2) up.push_back(n);
    }
    int64_t up_total = 0;
    for (int64_t n : up) up_total += n;
    if (ramp && !up.empty() && aligned >= 2 * up_total + 2 * window) {
      int64_t f = 0;
      for (int64_t n : up) { win_f0.push_back(f); win_nf.push_back(n); f += n; }
      const int64_t body_end = ramp_down ? aligned - up_total : aligned;
      while (f < body_end) {
        const int64_t n = std::min<int64_t>(window, body_end - f);
        win_f0.push_back(f); win_nf.push_back(n); f += n;
      }
      if (ramp_down)
        for (auto it = up.rbegin(); it != up.rend(); ++it) { win_f0.push_back(f); win_nf.push_back(*it); f += *it; }
    } else {
      for (int64_t f = 0; f < aligned; f += window) { win_f0.push_back(f); win_nf.push_back(std::min<int64_t>(window, aligned - f)); }
    }
    if (ragged > 0) { win_f0.push_back(aligned); win_nf.push_back(ragged); }
  }
  {
    int64_t longest = 1;
    for (int64_t n : win_nf) longest = std::max(longest, n);
    AF_HIP(e->supp.ensure_workspace(e->n_streams, (int)longest));
  }
  if (e->trace) {
    if (frames * e->n_streams > e->trace_capacity) {
      AF_HIP(hipDeviceSynchronize());
      if (e->d_trace) AF_HIP(hipFree(e->d_trace));
      e->d_trace = nullptr;
      AF_HIP(hipMalloc(&e->d_trace, sizeof(int32_t) * 2 * frames * e->n_streams));
      e->trace_capacity = frames * e->n_streams;
    }
    e->trace_frames = frames;
  }
  if (e->pipe.active) {
    // (a suppressor window enters the pipeline in pieces of the pipeline's own window length: the rings stay as small as
    // without the suppressor)
    const int64_t chain_tw = (int64_t)cb * std::max<int64_t>(1, 2880 / cb);
    if (int rc = stage_pipe_prepare(e, std::max<int64_t>(chain_tw, e->pipe.tw_max))) return rc;
    e->pipe.strip = e->host_params.flags & ~run.flags;  // what the pre-pass has taken over
    if (int rc = stage_chain_params(e, stream)) return rc;  // (everything but the EQ sections is read from here)
  }
  if (int rc = ensure_side_streams(e, stream)) return rc;
  static const bool split_synthesis = [] {  // AF_SYNTH_SPLIT=0: resynthesis + overlap-add stay behind the network on one stream
    const char *env = std::getenv("AF_SYNTH_SPLIT");
    return !env || std::atoi(env) != 0;
  }();
  const hipStream_t fin = split_synthesis ? e->fin_stream : nullptr;
  // The network kernel (a dependent chain of matrix instructions per 16 streams: long, and light on the chip) on a stream of its
  // own: behind the pitch spectra on ONE stream the pair took 3.0 of a window's 3.16 ms -- that stream was the pipeline's period.
  // MEASURED AND OFF: the eighth stream alone -- created, not even used -- takes the step from 172 to 184 ms, and with the network
  // on it 187-189 ms: HIP multiplexes a process's streams onto a handful of hardware queues, and one more stream makes two of
  // the pipeline's stages share a queue (false serialisation).  AF_RNN_STREAM=1 creates and uses it (A/B runs).
  static const bool rnn_own_stream = rnn_stream_wanted();
  static const bool rnn_on_caller = [] {  // AF_RNN_STREAM=2
    const char *env = std::getenv("AF_RNN_STREAM");
    return env && std::atoi(env) == 2;
  }();
  const hipStream_t syn = e->syn_stream ? e->syn_stream : stream;  // where the synthesis stage runs
  int64_t blocks_done = 0;
  std::vector<af::DiagWin> diag_wins;                 // the call's windows in the stage pipeline (one launch per step)
  const StagePlan diag_plan = stage_plan(run);
  static const bool eq_offload_env = [] {  // AF_EQ_OFFLOAD=0: the EQ stays inside the chain launches (A/B runs)
    const char *env = std::getenv("AF_EQ_OFFLOAD");
    return !env || std::atoi(env) != 0;
  }();
  const bool eq_offload = eq_offload_env && (e->kernel == AF_KERNEL_AUTO || e->kernel == AF_KERNEL_PHASED || e->kernel == AF_KERNEL_ROLES) &&
                          (e->ring_variant == 0 || e->ring_variant == 1604) && (run.flags & af::kFlagEq);
  bool eq_needs_chain_done = true;  // (the previous call's last chain launch has ended: the caller's stream waited for it)
  const bool auto_makeup_call = (run.flags & af::kFlagCompressor) && run.comp.auto_makeup_enabled;
  if (eq_offload && auto_makeup_call && !e->pipe.active) {
    // the systolic EQ kernel is then also the pre-pass of every window (it leaves the compressor-input block powers here)
    int64_t cap = e->block_power_capacity * (int64_t)sizeof(double);
    if (int rc = grow_device(e, reinterpret_cast<void **>(&e->d_block_power), &cap, rows * (int64_t)sizeof(double), stream)) return rc;
    e->block_power_capacity = cap / (int64_t)sizeof(double);
  }
  auto next_event = [&](hipEvent_t *out_ev) -> int { return engine_event(e, out_ev); };
  // The call's statistics rows are cleared ONCE, here (their fields are written by the kernels that own them).  Round 2 cleared
  // every window's rows in front of its EQ launch: a fill kernel on the suppressor's crowded CUs, 0.05-0.45 ms between the
  // window's overlap-add and its EQ -- on the path the first chain launches wait for.
  static const bool clear_per_window = [] {  // AF_STATS_CLEAR=window: round 2's per-window fills (A/B runs)
    const char *env = std::getenv("AF_STATS_CLEAR");
    return env && std::strcmp(env, "window") == 0;
  }();
  if (!clear_per_window) AF_HIP(hipMemsetAsync(e->d_stats, 0, sizeof(af::BlockStats) * rows, stream));
  // ---- ONE chain launch per call (round 3).  With the chain's CUs its own, the EQ on the suppressor's side and nothing that
  // changes the parameter block between windows, the token-ring kernel is launched once, for the whole call, before the first
  // window: a chunk waits until the counter `d_ready` covers its samples, and every window's EQ launch is followed by a
  // one-thread kernel that publishes the new count.  What that removes from the chain's stream: 53 dispatches and their
  // cross-stream dependencies (~0.1 ms each while six other queues are busy: the trace of tools/step_timeline.py), the
  // state planes' load and write-back per window, and the fill / drain of the 16-wave pipeline per launch.
  // AF_CHAIN_PERSISTENT=0 restores one launch per window (A/B runs).
  const bool persistent_env = one_launch_calls_enabled();
  bool persistent = persistent_env && eq_offload && !clear_per_window && e->partition_chain_cus > 0 && !e->pipe.active &&
                    !std::getenv("AF_DIAG_SKIP_CHAIN") && e->extra_presets.empty() && roles_mode() == 0 &&
                    (e->kernel == AF_KERNEL_AUTO || e->kernel == AF_KERNEL_PHASED) && layout == AF_LAYOUT_STREAM_MAJOR &&
                    !(run.flags & (af::kFlagDeesser | af::kFlagDcBlock | af::kFlagPreHighpass)) && run.n_eq_sections <= 16 &&
                    af::ring_kernel_dynamic_lds(run.n_eq_sections, run.lim.lookahead_samples, false) <= af::kMaxLdsBytes &&
                    (!auto_makeup_call || e->d_block_power != nullptr) && win_f0.size() >= 2;
  if (persistent) {
    if (!e->d_ready) AF_HIP(hipMalloc(&e->d_ready, sizeof(int64_t)));
    AF_HIP(hipMemsetAsync(e->d_ready, 0, sizeof(int64_t), stream));
  }
  {  // the side streams start after whatever the caller queued before this call
    hipEvent_t ev;
    if (int rc = next_event(&ev)) return rc;
    AF_HIP(hipEventRecord(ev, stream));
    AF_HIP(hipStreamWaitEvent(e->aux_stream, ev, 0));
    AF_HIP(hipStreamWaitEvent(e->pre_stream, ev, 0));
    AF_HIP(hipStreamWaitEvent(e->ana_stream, ev, 0));
    if (syn != stream) AF_HIP(hipStreamWaitEvent(syn, ev, 0));
    if (fin && fin != stream) AF_HIP(hipStreamWaitEvent(fin, ev, 0));
    if (e->eq_stream != stream) AF_HIP(hipStreamWaitEvent(e->eq_stream, ev, 0));
    if (e->lim_stream) AF_HIP(hipStreamWaitEvent(e->lim_stream, ev, 0));
    if (e->rnn_stream) AF_HIP(hipStreamWaitEvent(e->rnn_stream, ev, 0));
  }
  constexpr int kXh = af::SuppressorHost::kXhBuffers;
  // Pipeline depth.  The spectrum / record buffers of window w are free again when its synthesis has ended, and the synthesis
  // of w needs the analysis of w: with D buffer sets the loop analysis(w + D) <- synthesis(w) <- network(w) <- pitch spectra(w)
  // <- analysis(w) bounds the window period by (sum of those kernels) / D.  Round 2 ran D = 2 (the trace showed exactly that
  // period: 7.3 ms of dependent kernels per two windows); AF_SUPP_DEPTH=2 restores it for A/B runs.
  static const int depth = [] {
    const char *env = std::getenv("AF_SUPP_DEPTH");
    const int d = env ? std::atoi(env) : af::SuppressorHost::kSpecBuffers;
    return d < 2 ? 2 : (d > af::SuppressorHost::kSpecBuffers ? af::SuppressorHost::kSpecBuffers : d);
  }();
  const int ana_ahead = depth - 1, pre_ahead = depth;  // windows the analysis / the pre-pass run ahead of the synthesis
  auto window_args = [&](int64_t f0, int64_t nf, int64_t index) {
    af::SuppArgs sa{};
    sa.in = src;
    sa.in_stride = src_stride;
    sa.out = out;
    sa.xh = e->supp.d_xh + (size_t)(index % kXh) * e->supp.xh_floats;
    sa.X = e->supp.d_X + (size_t)(index % depth) * e->supp.ws_cells * af::kRnnFreq;
    sa.P = e->supp.d_P + (size_t)(index % depth) * e->supp.ws_cells * af::kRnnFreq;
    sa.ds = e->supp.d_ds;
    sa.rec = e->supp.d_rec + (size_t)(index % depth) * e->supp.ws_cells;
    sa.state = e->supp.d_state;
    sa.stream_stride = stream_stride;
    sa.n_streams = e->n_streams;
    sa.n_frames = (int)nf;
    sa.frame0 = f0;
    sa.strength = e->supp.strength;
    sa.smoothing_coeff = 1.0f - std::exp(-((480.0f / 48000.0f) / (15.0f / 1000.0f)));  // rnnoise.rs:45-51
    sa.raw_protocol = e->supp.raw_protocol ? 1 : 0;
    sa.front_clamp = (front_flags & af::kFlagInputClamp) ? 1 : 0;
    sa.front_dc = (front_flags & af::kFlagDcBlock) ? 1 : 0;
    sa.front_hp = (front_flags & af::kFlagPreHighpass) ? 1 : 0;
    sa.hp_b0 = run.pre_hp.b0; sa.hp_b1 = run.pre_hp.b1; sa.hp_b2 = run.pre_hp.b2;
    sa.hp_a1 = run.pre_hp.a1; sa.hp_a2 = run.pre_hp.a2;
    sa.chain_st64 = e->d_st64;
    sa.chain_st32 = e->d_st32;
    sa.f64_pre_z1 = af::kPreZ1;
    sa.f32_dc_x1 = af::kDcX1;
    if (index > 0) {  // history = tail of the previous window's buffer
      sa.xh_prev = e->supp.d_xh + (size_t)((index - 1) % kXh) * e->supp.xh_floats;
      sa.xh_prev_stride = af::kPitchBuf + win_nf[index - 1] * af::kRnnFrame;
    }
    return sa;
  };
  const int64_t n_windows = (int64_t)win_f0.size();
  std::vector<hipEvent_t> pre_done(n_windows), ana_done(n_windows), syn_done(n_windows);
  std::vector<hipEvent_t> rnn_done(n_windows);
  for (int64_t w = 0; w < n_windows; ++w) {
    if (int rc = next_event(&rnn_done[w])) return rc;
    if (int rc = next_event(&pre_done[w])) return rc;
    if (int rc = next_event(&ana_done[w])) return rc;
    if (int rc = next_event(&syn_done[w])) return rc;
  }
  // Stages are enqueued in pipeline order (the pre-pass two windows and the analysis one window ahead of the
  // synthesis), so that every event a stage waits on has been recorded before the wait is enqueued.
  auto enqueue_pre = [&](int64_t w) -> int {
    const int64_t f0 = win_f0[w], nf = win_nf[w];
    if (w >= kXh) AF_HIP(hipStreamWaitEvent(e->pre_stream, syn_done[w - kXh], 0));  // its model-input buffer is free
    AF_HIP(af::launch_suppressor_prefilter(window_args(f0, nf, w), e->pre_stream));
    AF_HIP(hipEventRecord(pre_done[w], e->pre_stream));
    return AF_OK;
  };
  auto enqueue_ana = [&](int64_t w) -> int {
    const int64_t f0 = win_f0[w], nf = win_nf[w];
    AF_HIP(hipStreamWaitEvent(e->ana_stream, pre_done[w], 0));
    if (w >= depth) AF_HIP(hipStreamWaitEvent(e->ana_stream, syn_done[w - depth], 0));  // its spectrum / record buffers are free
    static const bool order_pitch = [] {  // AF_ORDER_PITCH=1: hold the pitch search back until the previous window's network ran
      const char *env = std::getenv("AF_ORDER_PITCH");
      return env && std::atoi(env) != 0;  // off: it only moves the starvation to the resynthesis kernel (313 vs 296 ms)
    }();
    // ORDERING THAT IS LOAD-BEARING: the pitch search of window w + 1 and the pitch tracker of window w must stay on this ONE
    // stream, in this order.  The whitened pitch buffers (`d_ds`, 3.4 KB per frame and stream) are a single set: the tracker of
    // window w reads what the search of window w wrote, and nothing but stream order keeps the search of w + 1 from overwriting
    // it first.  (Round 2 moved the tracker to the pre-pass stream to shorten this stream: run-to-run bit-identity was lost --
    // that race.  Moving either kernel needs a second `d_ds` set and an event from the tracker to the next search.)  The
    // tracker also owns the stream's pitch state rows (last period / gain, cepstral ring, the 1728-sample history a NEW call's
    // first pre-pass reads: ordered through the caller's stream at the end of the call).
    AF_HIP(af::launch_suppressor_analysis(window_args(f0, nf, w), e->supp.tables, e->ana_stream,
                                          (order_pitch && w >= 1) ? rnn_done[w - 1] : nullptr));
    AF_HIP(hipEventRecord(ana_done[w], e->ana_stream));
    return AF_OK;
  };
  static const bool diag_no_chain_kernel = [] {  // AF_DIAG_NO_CHAIN_KERNEL=1 (timing experiments): everything but the chain launch
    const char *env = std::getenv("AF_DIAG_NO_CHAIN_KERNEL");
    return env && std::atoi(env) != 0;
  }();
  if (persistent && !diag_no_chain_kernel) {  // the call's one chain launch: resident on the chain's CUs from here on, following `d_ready`
    af::ChainParams run_p = run;
    run_p.flags = (run_p.flags & ~af::kFlagEq) | af::kFlagInputDone;  // (what every window's launch was given)
    const int64_t total = frames * af::kRnnFrame;
    if (int rc = launch_chain_segment(e, run_p, run_modified, out, out, total, stream_stride, layout, e->samples_processed, e->d_stats,
                                      e->has_evidence ? e->d_vad : nullptr, e->aux_stream, stream, /*stats_cleared=*/true,
                                      auto_makeup_call ? e->d_block_power : nullptr, e->d_ready))
      return rc;
    eq_needs_chain_done = false;  // (an event behind THIS launch would make the first EQ wait for the launch that waits for it)
  }
  for (int64_t w = 0; w < std::min<int64_t>(pre_ahead, n_windows); ++w)
    if (int rc = enqueue_pre(w)) return rc;
  for (int64_t w = 0; w < std::min<int64_t>(ana_ahead, n_windows); ++w)
    if (int rc = enqueue_ana(w)) return rc;
  for (int64_t w = 0; w < n_windows; ++w) {
    const int64_t f0 = win_f0[w], nf = win_nf[w];
    AF_HIP(hipStreamWaitEvent(syn, ana_done[w], 0));
    if (fin && fin != syn && w >= depth) AF_HIP(hipStreamWaitEvent(syn, syn_done[w - depth], 0));  // its pitch-spectrum buffer is free
    {
      hipStream_t net = nullptr;
      hipEvent_t spec_done = nullptr;
      if (rnn_own_stream && fin && fin != syn && e->rnn_stream) {
        net = e->rnn_stream;
        if (int rc = next_event(&spec_done)) return rc;
      } else if (rnn_on_caller && persistent && fin && fin != syn && syn != stream) {
        net = stream;  // the caller's stream: a queue the process has anyway, idle between the call's fork and its join
        if (int rc = next_event(&spec_done)) return rc;
      }
      AF_HIP(af::launch_suppressor_synthesis(window_args(f0, nf, w), e->supp.tables, e->supp.dw, syn, rnn_done[w], fin, net, spec_done));
    }
    if (e->trace) {  // the window's (silence, pitch index) decisions, before its record buffer is handed back to the analysis
      const af::SuppArgs sa = window_args(f0, nf, w);
      AF_HIP(hipMemcpy2DAsync(e->d_trace + 2 * f0 * e->n_streams, 2 * sizeof(int32_t),
                              reinterpret_cast<const char *>(sa.rec) + offsetof(af::SuppFrameRec, silence), sizeof(af::SuppFrameRec),
                              2 * sizeof(int32_t), (size_t)(nf * e->n_streams), hipMemcpyDeviceToDevice, (fin && fin != syn) ? fin : syn));
    }
    AF_HIP(hipEventRecord(syn_done[w], (fin && fin != syn) ? fin : syn));
    e->last_launches += 7;
    if (w + pre_ahead < n_windows)
      if (int rc = enqueue_pre(w + pre_ahead)) return rc;
    if (w + ana_ahead < n_windows)
      if (int rc = enqueue_ana(w + ana_ahead)) return rc;
    const int64_t seg0 = f0 * af::kRnnFrame, seg_n = nf * af::kRnnFrame;
    const double *vad = e->has_evidence ? e->d_vad + blocks_done * e->n_streams : nullptr;
    static const bool diag_skip_chain = std::getenv("AF_DIAG_SKIP_CHAIN") != nullptr;  // timing experiments only
    // ---- the window's EQ on the suppressor's side (af_eq_systolic.hip), when the chain's launch would be the plain
    // one-launch form of the token-ring kernel and no coefficient crossfade is running
    af::ChainParams run_w = run;
    bool eq_offloaded = false;
    double *power_w = nullptr;  // the window's block powers, when its systolic EQ launch was an auto-makeup pre-pass
    if (e->pipe.active && !diag_skip_chain) {
      // ---- the window's chain as one more step of the stage pipeline (af_stages.hip; small and medium batches): this window
      // enters (its EQ stage reads the overlap-add output), the windows before it move one stage on
      const hipStream_t ds = e->pipe.stream;
      const int64_t chain_tw = (int64_t)cb * std::max<int64_t>(1, 2880 / cb);
      AF_HIP(hipStreamWaitEvent(ds, syn_done[w], 0));
      if (clear_per_window)
        AF_HIP(hipMemsetAsync(e->d_stats + blocks_done * e->n_streams, 0, sizeof(af::BlockStats) * ((seg_n + cb - 1) / cb) * e->n_streams, ds));
      e->last_kernel_used = AF_KERNEL_STAGED;
      e->pipe.call_stride = stream_stride;
      int64_t sub_blocks = 0;
      for (int64_t off = 0; off < seg_n; off += chain_tw) {
        const int64_t n_sub = std::min<int64_t>(chain_tw, seg_n - off);
        af::DiagWin wd{};
        wd.n0 = e->samples_processed + seg0 + off;
        wd.n = n_sub;
        wd.stats = e->d_stats + (blocks_done + sub_blocks) * e->n_streams;
        wd.mk = e->pipe.d_mk + ((e->pipe.windows + (int64_t)diag_wins.size()) % af_engine::StagePipe::kMkSets) * e->pipe.mk_rows;
        wd.bp = e->pipe.d_bp + ((e->pipe.windows + (int64_t)diag_wins.size()) % af_engine::StagePipe::kBpSets) * e->pipe.mk_rows;
        wd.vad = vad ? vad + sub_blocks * e->n_streams : nullptr;
        wd.in = out + seg0 + off;
        wd.out = out + seg0 + off;
        bool crossfade = false;
        if (int rc2 = stage_diag_eq_params(e, ds, &crossfade, &wd.eq_slot)) return rc2;
        wd.eq_crossfade = crossfade ? 1 : 0;
        diag_wins.push_back(wd);
        if (int rc2 = stage_diag_step(e, run, diag_plan, diag_wins, (int64_t)diag_wins.size() - 1, ds)) return rc2;
        advance_crossfades(e, n_sub);
        sub_blocks += (n_sub + cb - 1) / cb;
      }
      run = e->host_params;  // crossfade bookkeeping may have moved on
      if (front_flags) run.flags &= ~(front | af::kFlagInputScrub);
      blocks_done += (seg_n + cb - 1) / cb;
      continue;
    }
    if (eq_offload && !diag_skip_chain) {
      const int n_presets = 1 + (int)e->extra_presets.size();
      bool ok = true, xf_w = false;
      std::vector<af::ChainParams> runs_eq((size_t)n_presets);
      for (int k = 0; k < n_presets && ok; ++k) {
        runs_eq[k] = preset_params(e, k);
        runs_eq[k].flags &= ~(e->host_params.flags & ~run.flags);  // what the pre-pass has taken over
        const af::ChainParams &hp = runs_eq[k];
        ok = !(hp.flags & af::kFlagDeesser) && hp.n_eq_sections <= 16 && !(hp.flags & (af::kFlagDcBlock | af::kFlagPreHighpass)) &&
             af::ring_kernel_dynamic_lds(hp.n_eq_sections, hp.lim.lookahead_samples, false) <= af::kMaxLdsBytes;
        // (a pending coefficient crossfade -- the 72 samples the legacy setters open a stream with -- runs in the systolic
        // kernel's general form; round 2 kept such windows' EQ inside the chain launch)
        for (int j = 0; j < hp.n_eq_sections; ++j) xf_w = xf_w || hp.eq[j].xf_remaining > 0;
      }
      if (ok) {
        // behind the window's overlap-add, beside the next window's synthesis (AF_EQ_ON_FIN=1: on the synthesis' own stream)
        static const bool eq_on_fin = [] {
          const char *env = std::getenv("AF_EQ_ON_FIN");
          return env && std::atoi(env) != 0;
        }();
        const hipStream_t es = (eq_on_fin && fin && fin != syn) ? fin : e->eq_stream;
        AF_HIP(hipStreamWaitEvent(es, syn_done[w], 0));
        if (!e->d_params_eq || e->eq_params_presets != n_presets) {
          if (e->d_params_eq) AF_HIP(hipFree(e->d_params_eq));
          e->d_params_eq = nullptr;
          AF_HIP(hipMalloc(&e->d_params_eq, sizeof(af::ChainParams) * n_presets * kEqParamSlots));  // (sized as the stage pipeline sizes it)
          e->eq_params_presets = n_presets;
          e->uploaded_eq.clear();
        }
        // AF_EQ_PARTS=2 (MEASURED, OFF): the window's EQ as TWO launches of the lane-per-stream kernel, sections [0, h) on the
        // caller's stream and [h, n) on the EQ stream, so that the second half of window w runs beside the first half of w + 1.
        // (With the chain launch left out of the step the EQ stream is the last to finish -- 52 x 3.2 ms = 166 ms -- hence the
        // attempt.)  Bit-identical; on a caller-owned stream 178.0 against 176.9-178.3 ms per step: nothing, and the caller's own
        // stream costs more than the default one (one more hardware queue).  On the legacy default stream it cannot run at all.
        static const bool eq_two_parts_env = [] {
          const char *env = std::getenv("AF_EQ_PARTS");
          return env && std::atoi(env) == 2;
        }();
        const bool two_parts = eq_two_parts_env && persistent && n_presets == 1 && !xf_w && (runs_eq[0].flags & af::kFlagEq) &&
                               runs_eq[0].n_eq_sections >= 2 && runs_eq[0].n_eq_sections <= 32 && stream != es && (stream_stride % 4) == 0 &&
                               // (not the legacy default stream -- or the per-thread one: work enqueued there waits for the other
                               // streams' earlier work, the resident chain launch included, which waits for this window; measured: the
                               // call runs into the launch's bound.  A caller on a stream of its own gets the two-part EQ.)
                               reinterpret_cast<uintptr_t>(stream) > 2 &&
                               (reinterpret_cast<uintptr_t>(out + seg0) & 15) == 0 && !std::getenv("AF_EQ_STREAM_OFF");
        if (e->uploaded_eq.size() != runs_eq.size() ||
            std::memcmp(e->uploaded_eq.data(), runs_eq.data(), sizeof(af::ChainParams) * runs_eq.size()) != 0) {
          e->uploaded_eq = runs_eq;
          // (always on the EQ stream, never the caller's: a copy on the legacy default stream waits for every other stream --
          // the resident chain launch included, which waits for this window: the call would run into the launch's bound)
          if (int rc2 = stage_upload(e, e->d_params_eq, runs_eq.data(), runs_eq.size(), es)) return rc2;
          if (stream != es) e->eq_params_on_es = true;
        }
        if (eq_needs_chain_done) {  // the previous window's EQ ran inside its chain launch: that launch owns the memories until it ends
          hipEvent_t chain_done;
          if (int rc2 = next_event(&chain_done)) return rc2;
          AF_HIP(hipEventRecord(chain_done, e->aux_stream));
          AF_HIP(hipStreamWaitEvent(es, chain_done, 0));
          eq_needs_chain_done = false;
        }
        static const bool eq_stream_with_power = [] {  // AF_EQ_STREAM_POWER=0: auto-makeup windows keep the systolic kernel
          const char *env = std::getenv("AF_EQ_STREAM_POWER");
          return !env || std::atoi(env) != 0;
        }();
        af::BlockStats *rows_w = e->d_stats + blocks_done * e->n_streams;
        if (clear_per_window) AF_HIP(hipMemsetAsync(rows_w, 0, sizeof(af::BlockStats) * ((seg_n + cb - 1) / cb) * e->n_streams, es));
        power_w = auto_makeup_call ? e->d_block_power + blocks_done * e->n_streams : nullptr;
        if (two_parts) {
          const int nsec = runs_eq[0].n_eq_sections, h = nsec / 2;
          if (e->eq_params_on_es) {  // (an upload the EQ stream made for an earlier window: the caller's stream reads the block now)
            hipEvent_t up;
            if (int rc2 = next_event(&up)) return rc2;
            AF_HIP(hipEventRecord(up, es));
            AF_HIP(hipStreamWaitEvent(stream, up, 0));
            e->eq_params_on_es = false;
          }
          AF_HIP(hipStreamWaitEvent(stream, syn_done[w], 0));
          AF_HIP(af::launch_eq_stream_part(e->d_params_eq, e->d_st64, out + seg0, out + seg0, rows_w, nullptr, 0, h, true, seg_n,
                                           stream_stride, e->n_streams, stream));
          hipEvent_t half;
          if (int rc2 = next_event(&half)) return rc2;
          AF_HIP(hipEventRecord(half, stream));
          AF_HIP(hipStreamWaitEvent(es, half, 0));
          AF_HIP(af::launch_eq_stream_part(e->d_params_eq, e->d_st64, out + seg0, out + seg0, power_w ? rows_w : nullptr, power_w, h, nsec - h,
                                           false, seg_n, stream_stride, e->n_streams, es));
          e->last_launches += 2;
          AF_HIP(af::launch_chain_publish_ready(e->d_ready, seg0 + seg_n, es));
          advance_crossfades(e, seg_n);
          run = e->host_params;
          if (front_flags) run.flags &= ~(front | af::kFlagInputScrub);
          blocks_done += (seg_n + cb - 1) / cb;
          continue;
        }
        AF_HIP(af::launch_eq_systolic(e->d_params_eq, e->extra_presets.empty() ? nullptr : e->d_group_preset, e->d_st64, out + seg0, out + seg0, nullptr, nullptr, 0, 0,
                                      rows_w, xf_w, seg_n, stream_stride, e->n_streams, es, power_w,
                                      // the lane-per-stream form where the suppressor's kernels want the issue slots and nothing waits
                                      // for the EQ's own latency (an auto-makeup window's block powers do): 184.5 -> 182.4 ms per step
                                      (n_presets == 1 && (!power_w || eq_stream_with_power) && (runs_eq[0].flags & af::kFlagEq)) ? runs_eq[0].n_eq_sections : -1));
        e->last_launches += 1;
        if (persistent) {  // the running chain launch picks the window up from here
          AF_HIP(af::launch_chain_publish_ready(e->d_ready, seg0 + seg_n, es));
          advance_crossfades(e, seg_n);  // (the one chain launch did not: the EQ's counters move window by window)
          run = e->host_params;
          if (front_flags) run.flags &= ~(front | af::kFlagInputScrub);
          blocks_done += (seg_n + cb - 1) / cb;
          continue;
        }
        hipEvent_t eq_done;
        if (int rc2 = next_event(&eq_done)) return rc2;
        AF_HIP(hipEventRecord(eq_done, es));
        AF_HIP(hipStreamWaitEvent(e->aux_stream, eq_done, 0));
        run_w.flags = (run_w.flags & ~af::kFlagEq) | af::kFlagInputDone;
        eq_offloaded = true;
      }
    }
    if (persistent) return fail(AF_ERR_BACKEND, "internal: a window of a one-launch call could not take the EQ on the suppressor's side");
    if (!eq_offloaded) {
      AF_HIP(hipStreamWaitEvent(e->aux_stream, syn_done[w], 0));
      eq_needs_chain_done = true;
    }
    int rc = AF_OK;
    if (!diag_skip_chain)
      rc = launch_chain_segment(e, run_w, run_modified, out + seg0, out + seg0, seg_n, stream_stride, layout,
                                e->samples_processed + seg0, e->d_stats + blocks_done * e->n_streams, vad, e->aux_stream, stream,
                                /*stats_cleared=*/!clear_per_window || eq_offloaded, power_w);
    if (rc) return rc;
    run = e->host_params;  // crossfade bookkeeping may have moved on
    if (front_flags) run.flags &= ~(front | af::kFlagInputScrub);
    blocks_done += (seg_n + cb - 1) / cb;
  }
  if (e->timing) AF_HIP(hipEventRecord(e->ev_mid, (fin && fin != syn) ? fin : syn));  // last suppressor kernel done
  if (syn != stream && n_windows > 0) AF_HIP(hipStreamWaitEvent(stream, syn_done[n_windows - 1], 0));
  {
    hipEvent_t ev;
    if (int rc = next_event(&ev)) return rc;
    AF_HIP(hipEventRecord(ev, e->aux_stream));
    AF_HIP(hipStreamWaitEvent(stream, ev, 0));
    if (e->lim_stream) {
      if (int rc = next_event(&ev)) return rc;
      AF_HIP(hipEventRecord(ev, e->lim_stream));
      AF_HIP(hipStreamWaitEvent(stream, ev, 0));
    }
  }
  if (e->pipe.active && !diag_wins.empty()) {
    const hipStream_t ds = e->pipe.stream;
    for (int64_t j = (int64_t)diag_wins.size(); j < (int64_t)diag_wins.size() + diag_plan.depth; ++j)  // the pipeline empties
      if (int rc = stage_diag_step(e, run, diag_plan, diag_wins, j, ds)) return rc;
    e->pipe.windows += (int64_t)diag_wins.size();
    hipEvent_t ev;
    if (int rc = next_event(&ev)) return rc;
    AF_HIP(hipEventRecord(ev, ds));
    AF_HIP(hipStreamWaitEvent(stream, ev, 0));
  }
  if (e->timing) AF_HIP(hipEventRecord(e->ev_stop, stream));
  e->samples_processed += n_samples;
  return AF_OK;
}

// host buffers in, host buffers out: `in` is [streams][n_in] (stream-major) or [n_in][streams] (time-major), `out` gets
// *n_out samples per stream at out_stride (stream-major) -- n_out differs from n_in only with the suppressor on
static int process_host_impl(af_engine *e, const float *in, int64_t n_in, float *out, int64_t out_stride, int32_t layout,
                             int64_t *n_out) {
  if (!e) return fail(AF_ERR_INVALID_ARGUMENT, "engine is null");
  if (n_in < 0) return fail(AF_ERR_INVALID_ARGUMENT, "n_samples must be >= 0");
  if (n_in > 0 && (!in || !out)) return fail(AF_ERR_INVALID_ARGUMENT, "audio pointers are null");
  if (int rc = ensure_started(e)) return rc;
  const int64_t B = e->n_streams;
  const bool stream_major = layout == AF_LAYOUT_STREAM_MAJOR;
  int64_t produced = n_in;
  if (e->supp.enabled) produced = ((e->pending + n_in) / af::kRnnFrame) * af::kRnnFrame;
  if (produced > 0 && !out) return fail(AF_ERR_INVALID_ARGUMENT, "audio pointers are null");
  if (stream_major && out_stride < produced)
    return fail(AF_ERR_INVALID_ARGUMENT, "this call completes %lld samples per stream (%d were pending) but `out` holds %lld: "
                "use af_engine_stream_host with a larger out_stride", (long long)produced, e->pending, (long long)out_stride);
  const int64_t io_stride = stream_major ? std::max<int64_t>(std::max(n_in, produced), 1) : B;
  const int64_t total = stream_major ? io_stride * B : n_in * B;
  if (total > e->io_capacity) {
    if (e->d_io) AF_HIP(hipFree(e->d_io));
    e->d_io = nullptr;
    AF_HIP(hipMalloc(&e->d_io, sizeof(float) * total));
    e->io_capacity = total;
  }
  if (n_in > 0) {
    if (stream_major)
      AF_HIP(hipMemcpy2D(e->d_io, sizeof(float) * io_stride, in, sizeof(float) * n_in, sizeof(float) * n_in, B, hipMemcpyHostToDevice));
    else
      AF_HIP(hipMemcpy(e->d_io, in, sizeof(float) * total, hipMemcpyHostToDevice));
  }
  if (int rc = af_engine_process_device(e, e->d_io, e->d_io, n_in, io_stride, layout, nullptr)) return rc;
  AF_HIP(hipStreamSynchronize(nullptr));
  if (produced > 0) {
    if (stream_major)
      AF_HIP(hipMemcpy2D(out, sizeof(float) * out_stride, e->d_io, sizeof(float) * io_stride, sizeof(float) * produced, B, hipMemcpyDeviceToHost));
    else
      AF_HIP(hipMemcpy(out, e->d_io, sizeof(float) * total, hipMemcpyDeviceToHost));
  }
  if (n_out) *n_out = produced;
  return check_device_status(e);
}

int af_engine_process_host(af_engine *e, const float *in, float *out, int64_t n_samples, int32_t layout) {
  if (layout != AF_LAYOUT_STREAM_MAJOR && layout != AF_LAYOUT_TIME_MAJOR)
    return fail(AF_ERR_INVALID_ARGUMENT, "unknown layout %d", layout);
  return process_host_impl(e, in, n_samples, out, layout == AF_LAYOUT_STREAM_MAJOR ? n_samples : (e ? e->n_streams : 0), layout, nullptr);
}

int af_engine_stream_host(af_engine *e, const float *in, int64_t n_in, float *out, int64_t out_stride, int64_t *n_out) {
  if (n_out) *n_out = 0;
  return process_host_impl(e, in, n_in, out, out_stride, AF_LAYOUT_STREAM_MAJOR, n_out);
}

int64_t af_engine_pending_input(const af_engine *e) { return e ? e->pending : 0; }
int64_t af_engine_last_output_samples(const af_engine *e) { return e ? e->last_output_samples : 0; }

// test tap: RNNoiseProcessor::scale_sample_for_model (rnnoise.rs:89-111) as the pre-pass kernel evaluates it
int af_suppressor_debug_scale_for_model(const float *in, float *out, int64_t n, int32_t device) {
  if ((!in || !out) && n > 0) return fail(AF_ERR_INVALID_ARGUMENT, "null argument");
  if (n <= 0) return AF_OK;
  AF_HIP(hipSetDevice(device));
  float *d = nullptr;
  AF_HIP(hipMalloc(&d, sizeof(float) * n));
  hipError_t err = hipMemcpy(d, in, sizeof(float) * n, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = af::launch_scale_probe(d, d, n, nullptr);
  if (err == hipSuccess) err = hipMemcpy(out, d, sizeof(float) * n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (err != hipSuccess) return fail(AF_ERR_BACKEND, "scale probe failed: %s", hipGetErrorString(err));
  return AF_OK;
}

int af_suppressor_set_trace_enabled(af_engine *e, int32_t on) {
  if (!e) return fail(AF_ERR_INVALID_ARGUMENT, "engine is null");
  e->trace = on != 0;
  return AF_OK;
}
int64_t af_suppressor_trace_frames(const af_engine *e) { return e ? e->trace_frames : 0; }
int af_suppressor_read_trace(af_engine *e, int32_t *out, int64_t capacity_frames) {
  if (!e || !out) return fail(AF_ERR_INVALID_ARGUMENT, "null argument");
  if (capacity_frames < e->trace_frames)
    return fail(AF_ERR_INVALID_ARGUMENT, "capacity %lld < %lld frames", (long long)capacity_frames, (long long)e->trace_frames);
  if (e->trace_frames == 0) return AF_OK;
  AF_HIP(hipSetDevice(e->device));
  AF_HIP(hipStreamSynchronize(e->last_stream));
  AF_HIP(hipMemcpy(out, e->d_trace, sizeof(int32_t) * 2 * e->trace_frames * e->n_streams, hipMemcpyDeviceToHost));
  return AF_OK;
}

int af_engine_synchronize(af_engine *e) {
  if (!e) return fail(AF_ERR_INVALID_ARGUMENT, "engine is null");
  if (!e->started) return AF_OK;
  AF_HIP(hipSetDevice(e->device));
  AF_HIP(hipStreamSynchronize(e->last_stream));
  return check_device_status(e);
}

int64_t af_engine_last_block_count(const af_engine *e) { return e ? e->last_blocks : 0; }
int64_t af_engine_samples_processed(const af_engine *e) { return e ? e->samples_processed : 0; }

int af_engine_read_block_stats(af_engine *e, af_block_stats *out, int64_t capacity) {
  if (!e || !out) return fail(AF_ERR_INVALID_ARGUMENT, "null argument");
  const int64_t rows = e->last_blocks * e->n_streams;
  if (capacity < rows) return fail(AF_ERR_INVALID_ARGUMENT, "capacity %lld < %lld rows", (long long)capacity, (long long)rows);
  if (rows == 0) return AF_OK;
  AF_HIP(hipSetDevice(e->device));
  AF_HIP(hipStreamSynchronize(e->last_stream));
  AF_HIP(hipMemcpy(out, e->d_stats, sizeof(af::BlockStats) * rows, hipMemcpyDeviceToHost));
  return check_device_status(e);
}

int af_engine_last_kernel_ms(af_engine *e, double *ms, int32_t *launches) {
  if (!e || !ms) return fail(AF_ERR_INVALID_ARGUMENT, "null argument");
  *ms = 0.0;
  if (launches) *launches = e->last_launches;
  if (!e->timing || !e->ev_start || e->last_launches == 0) return AF_OK;
  AF_HIP(hipSetDevice(e->device));
  AF_HIP(hipEventSynchronize(e->ev_stop));
  float t = 0.0f;
  AF_HIP(hipEventElapsedTime(&t, e->ev_start, e->ev_stop));
  *ms = (double)t;
  return AF_OK;
}

int af_engine_last_stage_ms(af_engine *e, double *suppressor_ms, double *chain_ms) {
  if (!e || !suppressor_ms || !chain_ms) return fail(AF_ERR_INVALID_ARGUMENT, "null argument");
  *suppressor_ms = *chain_ms = 0.0;
  if (!e->timing || !e->ev_start || e->last_launches == 0) return AF_OK;
  AF_HIP(hipSetDevice(e->device));
  AF_HIP(hipEventSynchronize(e->ev_stop));
  float t = 0.0f;
  AF_HIP(hipEventElapsedTime(&t, e->ev_start, e->ev_mid));
  *suppressor_ms = (double)t;
  double chain = 0.0;
  for (auto &pr : e->chain_ms_events) {
    AF_HIP(hipEventSynchronize(pr.second));
    AF_HIP(hipEventElapsedTime(&t, pr.first, pr.second));
    chain += (double)t;
  }
  *chain_ms = chain;  // summed over the chain launches of the call (they may overlap suppressor kernels)
  return AF_OK;
}

int af_engine_last_chain_launch_ms(af_engine *e, double *first_ms, double *tail_ms, int32_t *segments) {
  if (!e || !first_ms || !tail_ms) return fail(AF_ERR_INVALID_ARGUMENT, "null argument");
  *first_ms = *tail_ms = 0.0;
  if (segments) *segments = (int32_t)e->chain_ms_events.size();
  if (!e->timing || !e->ev_start || e->last_launches == 0) return AF_OK;
  AF_HIP(hipSetDevice(e->device));
  float t = 0.0f;
  for (auto &pr : e->chain_ms_events) {
    AF_HIP(hipEventSynchronize(pr.second));
    AF_HIP(hipEventElapsedTime(&t, pr.first, pr.second));
    *first_ms += (double)t;
  }
  return AF_OK;
}

// ---- stateless helpers -----------------------------------------------------------------
int af_eq_magnitude_response(const double *freqs, size_t n, const double bands[10][3], double sample_rate,
                             double *out_db) {  // lib.rs:99-150
  if (!std::isfinite(sample_rate) || sample_rate <= 0.0)
    return fail(AF_ERR_INVALID_ARGUMENT, "sample_rate must be finite and positive");
  if (!bands || (!freqs && n) || (!out_db && n)) return fail(AF_ERR_INVALID_ARGUMENT, "null argument");
  const double nyquist = sample_rate / 2.0;
  for (int i = 0; i < af::kNumBands; ++i) {
    const double f = bands[i][0], g = bands[i][1], q = bands[i][2];
    if (!std::isfinite(f) || f <= 0.0 || f >= nyquist)
      return fail(AF_ERR_INVALID_ARGUMENT, "band %d frequency must be between 0 Hz and Nyquist", i);
    if (!std::isfinite(g)) return fail(AF_ERR_INVALID_ARGUMENT, "band %d gain must be finite", i);
    if (!std::isfinite(q) || q <= 0.0) return fail(AF_ERR_INVALID_ARGUMENT, "band %d Q must be finite and positive", i);
  }
  for (size_t i = 0; i < n; ++i)
    if (!std::isfinite(freqs[i]) || freqs[i] < 0.0 || freqs[i] > nyquist)
      return fail(AF_ERR_INVALID_ARGUMENT, "response frequencies must be finite and between 0 Hz and Nyquist");
  af::EqProto eq(sample_rate);
  for (int i = 0; i < af::kNumBands; ++i) {
    eq.set_band_frequency(i, bands[i][0]);
    eq.set_band_gain(i, bands[i][1]);
    eq.set_band_q(i, bands[i][2]);
  }
  for (size_t i = 0; i < n; ++i) out_db[i] = eq.magnitude_db(freqs[i]);
  return AF_OK;
}

int af_eq_magnitude_response_v2(const double *freqs, size_t n, const af_eq_band_config bands[10], double sample_rate,
                                double *out_db) {  // lib.rs:152-212
  if (!std::isfinite(sample_rate) || sample_rate <= 0.0)
    return fail(AF_ERR_INVALID_ARGUMENT, "sample_rate must be finite and positive");
  if (!bands || (!freqs && n) || (!out_db && n)) return fail(AF_ERR_INVALID_ARGUMENT, "null argument");
  for (int i = 0; i < af::kNumBands; ++i)
    if (int rc = af_eq_band_config_validate(&bands[i], i, sample_rate)) return rc;
  const double nyquist = sample_rate / 2.0;
  for (size_t i = 0; i < n; ++i)
    if (!std::isfinite(freqs[i]) || freqs[i] < 0.0 || freqs[i] > nyquist)
      return fail(AF_ERR_INVALID_ARGUMENT, "response frequencies must be finite and between 0 Hz and Nyquist");
  af::EqProto eq(sample_rate);
  for (int i = 0; i < af::kNumBands; ++i) eq.set_band_config(i, to_cfg(bands[i]));
  for (size_t i = 0; i < n; ++i) out_db[i] = eq.magnitude_db(freqs[i]);
  return AF_OK;
}

int af_engine_eq_magnitude_response(const af_engine *e, const double *freqs, size_t n, double *out_db) {
  if (!e || (!freqs && n) || (!out_db && n)) return fail(AF_ERR_INVALID_ARGUMENT, "null argument");
  for (size_t i = 0; i < n; ++i) out_db[i] = cur(e).eq.magnitude_db(freqs[i]);
  return AF_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// Product resampler (rust-core/src/audio/processor/resampling.rs:140-261)
struct af_resampler {
  af::ResamplePlan plan;
  int device = 0;
  std::vector<af::ResamplePos> pos;
  int64_t planned_n_in = -1, planned_n_out = 0, planned_blocks = 0, uploaded_n_in = -1;
  double *d_table = nullptr;
  af::ResamplePos *d_pos = nullptr;
  int64_t pos_capacity = 0;
  double *d_in = nullptr, *d_out = nullptr;
  int64_t in_capacity = 0, out_capacity = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  int variant = 0;  // 0: matrix-core kernel (64 streams per workgroup) when the shape allows; AF_RESAMPLER_VARIANT=valu -> 1: vector kernel, =mfma32 -> 2: matrix-core kernel with 32 streams per workgroup
};

namespace {
// host only: replay the reference's chunk loop for n_in frames
int resampler_plan(af_resampler *r, int64_t n_in) {
  if (r->planned_n_in == n_in) return AF_OK;
  r->planned_n_out = r->plan.positions(n_in, r->pos, &r->planned_blocks);
  r->planned_n_in = n_in;
  r->uploaded_n_in = -1;
  return AF_OK;
}
// device side of the plan: coefficient table (once) and the position records of the current plan
int resampler_upload(af_resampler *r) {
  AF_HIP(hipSetDevice(r->device));
  if (!r->d_table) {
    AF_HIP(hipMalloc(&r->d_table, sizeof(double) * r->plan.table.size()));
    AF_HIP(hipMemcpy(r->d_table, r->plan.table.data(), sizeof(double) * r->plan.table.size(), hipMemcpyHostToDevice));
  }
  if (r->uploaded_n_in == r->planned_n_in) return AF_OK;
  if (r->planned_n_out > r->pos_capacity) {
    if (r->d_pos) AF_HIP(hipFree(r->d_pos));
    r->d_pos = nullptr;
    AF_HIP(hipMalloc(&r->d_pos, sizeof(af::ResamplePos) * r->planned_n_out));
    r->pos_capacity = r->planned_n_out;
  }
  if (r->planned_n_out > 0)
    AF_HIP(hipMemcpy(r->d_pos, r->pos.data(), sizeof(af::ResamplePos) * r->planned_n_out, hipMemcpyHostToDevice));
  r->uploaded_n_in = r->planned_n_in;
  return AF_OK;
}
}  // namespace

extern "C" {

int af_resampler_calculate_cutoff(int32_t sinc_len, int32_t window, float *out) {
  if (!out) return fail(AF_ERR_INVALID_ARGUMENT, "out is null");
  if (window < 0 || window > af::kWinHann2) return fail(AF_ERR_INVALID_ARGUMENT, "unsupported resampler window %d", window);
  *out = af::resample_calculate_cutoff(sinc_len, window);
  return AF_OK;
}

int af_resampler_create(uint32_t input_rate, uint32_t output_rate, int64_t chunk_size, int32_t sinc_len, int32_t window,
                        int32_t device, af_resampler **out) {
  if (!out) return fail(AF_ERR_INVALID_ARGUMENT, "out is null");
  *out = nullptr;
  // the argument checks of simulate_product_resampler, resampling.rs:187-214
  if (input_rate == 0 || output_rate == 0) return fail(AF_ERR_INVALID_ARGUMENT, "sample rates must be positive");
  if (chunk_size < 1 || chunk_size > 1024) return fail(AF_ERR_INVALID_ARGUMENT, "chunk_size must be between 1 and 1024");
  if (sinc_len < 32 || sinc_len > 2048 || (sinc_len & (sinc_len - 1)) != 0)
    return fail(AF_ERR_INVALID_ARGUMENT, "sinc_len must be a power of two between 32 and 2048");
  if (window < 0 || window > af::kWinHann2) return fail(AF_ERR_INVALID_ARGUMENT, "unsupported resampler window %d", window);
  if (device < 0) return fail(AF_ERR_INVALID_ARGUMENT, "device must be >= 0");
  const double ratio = (double)output_rate / (double)input_rate;
  if (chunk_size <= (int64_t)sinc_len + 1 + (int64_t)std::ceil(1.0 / ratio))
    return fail(AF_ERR_UNSUPPORTED, "chunk_size %lld is too short for sinc_len %d: the reference's chunk loop would produce no frames",
                (long long)chunk_size, sinc_len);
  if (af::resample_segment_outputs(ratio, sinc_len) == 0)
    return fail(AF_ERR_UNSUPPORTED, "sinc_len %d at ratio %.4f needs a longer input span than the LDS tile holds", sinc_len, ratio);
  af_resampler *r = new af_resampler();
  r->device = device;
  r->plan.build(input_rate, output_rate, chunk_size, sinc_len, window);
  if (const char *env = std::getenv("AF_RESAMPLER_VARIANT")) r->variant = std::strcmp(env, "valu") == 0 ? 1 : (std::strcmp(env, "mfma32") == 0 ? 2 : 0);
  *out = r;
  return AF_OK;
}

void af_resampler_destroy(af_resampler *r) {
  if (!r) return;
  if (r->d_table || r->d_pos || r->d_in || r->d_out) {
    (void)hipSetDevice(r->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(r->d_table);
    (void)hipFree(r->d_pos);
    (void)hipFree(r->d_in);
    (void)hipFree(r->d_out);
  }
  if (r->ev0) (void)hipEventDestroy(r->ev0);
  if (r->ev1) (void)hipEventDestroy(r->ev1);
  delete r;
}

int af_resampler_output_delay(const af_resampler *r) { return r ? r->plan.output_delay() : 0; }
int64_t af_resampler_expected_frames(const af_resampler *r, int64_t n_in) { return r ? r->plan.expected_frames(n_in) : 0; }
int af_resampler_sinc_len(const af_resampler *r) { return r ? r->plan.sinc_len : 0; }

int af_resampler_copy_sinc_table(const af_resampler *r, double *out) {
  if (!r || !out) return fail(AF_ERR_INVALID_ARGUMENT, "null argument");
  const int stride = r->plan.row_stride();
  for (int row = 0; row < af::kResampleOversampling; ++row)
    std::memcpy(out + (size_t)row * r->plan.sinc_len, r->plan.table.data() + (size_t)row * stride + af::kResampleTablePad,
                sizeof(double) * r->plan.sinc_len);
  return AF_OK;
}

int af_resampler_plan(af_resampler *r, int64_t n_in, int64_t *n_out, int64_t *blocks) {
  if (!r) return fail(AF_ERR_INVALID_ARGUMENT, "resampler is null");
  if (n_in < 0) return fail(AF_ERR_INVALID_ARGUMENT, "n_in must be >= 0");
  if (int rc = resampler_plan(r, n_in)) return rc;
  if (n_out) *n_out = r->planned_n_out;
  if (blocks) *blocks = r->planned_blocks;
  return AF_OK;
}

int af_resampler_process_device(af_resampler *r, const double *d_in, double *d_out, int64_t n_in, int32_t n_streams,
                                int64_t in_stride, int64_t out_stride, void *stream) {
  if (!r) return fail(AF_ERR_INVALID_ARGUMENT, "resampler is null");
  if (n_streams <= 0) return fail(AF_ERR_INVALID_ARGUMENT, "n_streams must be positive");
  if (n_in < 0 || in_stride < n_in) return fail(AF_ERR_INVALID_ARGUMENT, "in_stride must cover n_in frames");
  if (int rc = resampler_plan(r, n_in)) return rc;
  if (out_stride < r->planned_n_out) return fail(AF_ERR_INVALID_ARGUMENT, "out_stride must cover the %lld planned output frames", (long long)r->planned_n_out);
  if ((!d_in && n_in > 0) || !d_out) return fail(AF_ERR_INVALID_ARGUMENT, "null device buffer");
  if (int rc = resampler_upload(r)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!r->ev0) {
    AF_HIP(hipEventCreate(&r->ev0));
    AF_HIP(hipEventCreate(&r->ev1));
  }
  AF_HIP(hipEventRecord(r->ev0, s));
  AF_HIP(af::launch_resample(d_in, d_out, r->d_pos, r->d_table, n_in, r->planned_n_out, in_stride, out_stride, n_streams,
                             r->plan.sinc_len, r->plan.ratio, r->variant, s));
  AF_HIP(hipEventRecord(r->ev1, s));
  r->timed = true;
  return AF_OK;
}

int af_resampler_process_host(af_resampler *r, const double *in, double *out, int64_t n_in, int32_t n_streams,
                              int64_t in_stride, int64_t out_stride) {
  if (!r) return fail(AF_ERR_INVALID_ARGUMENT, "resampler is null");
  if (n_streams <= 0) return fail(AF_ERR_INVALID_ARGUMENT, "n_streams must be positive");
  if ((!in && n_in > 0) || !out) return fail(AF_ERR_INVALID_ARGUMENT, "null buffer");
  if (n_in < 0 || in_stride < n_in) return fail(AF_ERR_INVALID_ARGUMENT, "in_stride must cover n_in frames");
  for (int64_t s = 0; s < n_streams; ++s)
    for (int64_t i = 0; i < n_in; ++i)
      if (!std::isfinite(in[s * in_stride + i])) return fail(AF_ERR_NON_FINITE, "samples must be finite");
  if (int rc = resampler_plan(r, n_in)) return rc;
  const int64_t n_out = r->planned_n_out;
  if (out_stride < n_out) return fail(AF_ERR_INVALID_ARGUMENT, "out_stride must cover the %lld planned output frames", (long long)n_out);
  const int64_t need_in = std::max<int64_t>(1, (int64_t)n_streams * n_in), need_out = std::max<int64_t>(1, (int64_t)n_streams * n_out);
  if (need_in > r->in_capacity) {
    if (r->d_in) AF_HIP(hipFree(r->d_in));
    r->d_in = nullptr;
    AF_HIP(hipMalloc(&r->d_in, sizeof(double) * need_in));
    r->in_capacity = need_in;
  }
  if (need_out > r->out_capacity) {
    if (r->d_out) AF_HIP(hipFree(r->d_out));
    r->d_out = nullptr;
    AF_HIP(hipMalloc(&r->d_out, sizeof(double) * need_out));
    r->out_capacity = need_out;
  }
  if (n_in > 0)
    AF_HIP(hipMemcpy2D(r->d_in, sizeof(double) * n_in, in, sizeof(double) * in_stride, sizeof(double) * n_in, n_streams, hipMemcpyHostToDevice));
  if (int rc = af_resampler_process_device(r, r->d_in, r->d_out, n_in, n_streams, n_in > 0 ? n_in : 1, n_out, nullptr)) return rc;
  AF_HIP(hipStreamSynchronize(nullptr));
  if (n_out > 0)
    AF_HIP(hipMemcpy2D(out, sizeof(double) * out_stride, r->d_out, sizeof(double) * n_out, sizeof(double) * n_out, n_streams, hipMemcpyDeviceToHost));
  return AF_OK;
}

int af_resampler_last_kernel_ms(af_resampler *r, double *ms) {
  if (!r || !ms) return fail(AF_ERR_INVALID_ARGUMENT, "null argument");
  *ms = 0.0;
  if (!r->timed) return AF_OK;
  AF_HIP(hipEventSynchronize(r->ev1));
  float t = 0.0f;
  AF_HIP(hipEventElapsedTime(&t, r->ev0, r->ev1));
  *ms = t;
  return AF_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// measure_integrated_loudness (lib.rs:290-298 over dsp/loudness.rs:43-83)
namespace {
// ebur128 `Mode::HISTOGRAM`: 1000 bins of 0.1 LU from -70 LUFS; blocks are represented by their bin's centre
double hist_energy(int i) { return std::pow(10.0, ((double)i / 10.0 - 69.95 + 0.691) / 10.0); }
double hist_boundary(int i) { return std::pow(10.0, ((double)i / 10.0 - 70.0 + 0.691) / 10.0); }
size_t find_histogram_index(double energy) {
  size_t lo = 0, hi = 1000;
  do {
    const size_t mid = (lo + hi) / 2;
    if (energy >= hist_boundary((int)mid)) lo = mid; else hi = mid;
  } while (hi - lo != 1);
  return lo;
}
// 400 ms blocks every 100 ms, absolute gate -70 LUFS, relative gate -10 LU; false when nothing passes the gates
bool gated_loudness(const double *part, int64_t n100, int64_t s100, double *lufs) {
  std::vector<uint64_t> counts(1000, 0);
  const double frames = (double)(s100 * 4);
  for (int64_t b = 0; b + 4 <= n100; ++b) {
    const double energy = (((part[b] + part[b + 1]) + part[b + 2]) + part[b + 3]) / frames;
    if (energy >= hist_boundary(0)) counts[find_histogram_index(energy)]++;
  }
  double rel = 0.0;
  uint64_t above = 0;
  for (int i = 0; i < 1000; ++i) { rel += (double)counts[i] * hist_energy(i); above += counts[i]; }
  if (!above) return false;
  rel /= (double)above;
  rel *= std::pow(10.0, -10.0 / 10.0);
  size_t start;
  if (rel < hist_boundary(0)) start = 0;
  else { start = find_histogram_index(rel); if (rel > hist_energy((int)start)) ++start; }
  double gated = 0.0;
  above = 0;
  for (size_t i = start; i < 1000; ++i) { gated += (double)counts[i] * hist_energy((int)i); above += counts[i]; }
  if (!above) return false;
  gated /= (double)above;
  *lufs = 10.0 * (std::log(gated) / std::log(10.0)) - 0.691;
  return std::isfinite(*lufs);
}
}  // namespace

extern "C" {

int af_measure_integrated_loudness_device(const float *d_audio, int64_t n_samples, int32_t n_streams, int64_t stream_stride,
                                          uint32_t sample_rate, int32_t device, double *lufs, int32_t *status) {
  // validate_sample_rate, loudness.rs:36-41
  static const uint32_t rates[] = {8000, 16000, 32000, 44100, 48000, 88200, 96000};
  bool rate_ok = false;
  for (uint32_t r : rates) rate_ok |= r == sample_rate;
  if (!rate_ok) return fail(AF_ERR_INVALID_ARGUMENT, "Invalid sample rate: %u", sample_rate);
  if (n_samples <= 0) return fail(AF_ERR_INVALID_ARGUMENT, "Invalid audio: at least one sample is required");
  if (n_streams <= 0 || !d_audio || !lufs) return fail(AF_ERR_INVALID_ARGUMENT, "null or empty batch");
  if (stream_stride < n_samples) return fail(AF_ERR_INVALID_ARGUMENT, "stream_stride must cover n_samples");
  AF_HIP(hipSetDevice(device));
  double b[5], a[5];
  af::kweighting_design((double)sample_rate, b, a);
  const int64_t s100 = ((int64_t)sample_rate + 5) / 10, n100 = n_samples / s100;
  double *d_part = nullptr;
  int32_t *d_bad = nullptr;
  AF_HIP(hipMalloc(&d_part, sizeof(double) * std::max<int64_t>(1, n100) * n_streams));
  AF_HIP(hipMalloc(&d_bad, sizeof(int32_t) * n_streams));
  hipError_t err = af::launch_kweight_energy(d_audio, d_part, d_bad, b, a, n_samples, stream_stride, n100, n_streams, (int32_t)s100, nullptr);
  std::vector<double> part((size_t)std::max<int64_t>(1, n100) * n_streams);
  std::vector<int32_t> bad(n_streams);
  if (err == hipSuccess) err = hipMemcpy(part.data(), d_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost);
  if (err == hipSuccess) err = hipMemcpy(bad.data(), d_bad, sizeof(int32_t) * n_streams, hipMemcpyDeviceToHost);
  (void)hipFree(d_part);
  (void)hipFree(d_bad);
  if (err != hipSuccess) return fail(AF_ERR_BACKEND, "integrated loudness failed: %s", hipGetErrorString(err));
  int worst = AF_OK;
  for (int32_t s = 0; s < n_streams; ++s) {
    int st = AF_OK;
    double v = -HUGE_VAL;
    if (bad[s]) st = AF_ERR_NON_FINITE;
    else if (!gated_loudness(part.data() + (size_t)s * std::max<int64_t>(1, n100), n100, s100, &v)) st = AF_ERR_UNSUPPORTED;
    lufs[s] = v;
    if (status) status[s] = st;
    if (st != AF_OK && worst == AF_OK) worst = st;
  }
  if (worst == AF_ERR_NON_FINITE) return fail(worst, "Invalid audio: samples must be finite");
  if (worst == AF_ERR_UNSUPPORTED)
    return fail(AF_ERR_INVALID_ARGUMENT, "Loudness measurement failed: audio did not produce a finite gated loudness");
  return AF_OK;
}

int af_measure_integrated_loudness_host(const float *audio, int64_t n_samples, int32_t n_streams, int64_t stream_stride,
                                        uint32_t sample_rate, int32_t device, double *lufs, int32_t *status) {
  if (!audio && n_samples > 0) return fail(AF_ERR_INVALID_ARGUMENT, "audio is null");
  if (n_samples <= 0) return fail(AF_ERR_INVALID_ARGUMENT, "Invalid audio: at least one sample is required");
  if (n_streams <= 0) return fail(AF_ERR_INVALID_ARGUMENT, "n_streams must be positive");
  AF_HIP(hipSetDevice(device));
  float *d_audio = nullptr;
  AF_HIP(hipMalloc(&d_audio, sizeof(float) * (size_t)n_samples * n_streams));
  hipError_t err = hipMemcpy2D(d_audio, sizeof(float) * n_samples, audio, sizeof(float) * stream_stride, sizeof(float) * n_samples,
                               n_streams, hipMemcpyHostToDevice);
  int rc = AF_OK;
  if (err != hipSuccess) rc = fail(AF_ERR_BACKEND, "hipMemcpy2D failed: %s", hipGetErrorString(err));
  else rc = af_measure_integrated_loudness_device(d_audio, n_samples, n_streams, n_samples, sample_rate, device, lufs, status);
  (void)hipFree(d_audio);
  return rc;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// Noise gate, expander path (dsp/gate.rs:626-637), as simulate_gate_suppressor_order drives it
namespace af {
struct GateArgs {
  const float *in;
  float *out;
  float *gain_trace;
  uint64_t *chatter;
  double *state;
  double threshold_db, attack_coeff, release_coeff, rms_coeff;
  int64_t n_samples, stride;
  int32_t n_streams, block, vad_mode;
  int32_t hold_samples, window_samples, cooldown_samples, relax_samples;
};
hipError_t launch_gate(const GateArgs &a, hipStream_t stream);
}  // namespace af

extern "C" {

int af_gate_process_host(const float *in, float *out, int64_t n_samples, int32_t n_streams, int64_t stream_stride,
                         double threshold_db, double attack_ms, double release_ms, double sample_rate, int32_t vad_mode,
                         int32_t trace_block, float *gain_trace, uint64_t *chatter_events, int32_t device) {
  if (!in || !out) return fail(AF_ERR_INVALID_ARGUMENT, "audio pointers are null");
  if (n_samples < 0 || n_streams <= 0 || stream_stride < n_samples) return fail(AF_ERR_INVALID_ARGUMENT, "bad batch shape");
  if (!std::isfinite(sample_rate) || sample_rate <= 0.0) return fail(AF_ERR_INVALID_ARGUMENT, "sample_rate must be positive and finite");
  if (trace_block <= 0) trace_block = 480;
  AF_HIP(hipSetDevice(device));
  const int64_t blocks = (n_samples + trace_block - 1) / trace_block;
  float *d_in = nullptr, *d_trace = nullptr;
  uint64_t *d_chatter = nullptr;
  const size_t audio_bytes = sizeof(float) * (size_t)std::max<int64_t>(1, n_samples) * n_streams;
  AF_HIP(hipMalloc(&d_in, audio_bytes));
  AF_HIP(hipMalloc(&d_trace, sizeof(float) * (size_t)std::max<int64_t>(1, blocks) * n_streams));
  AF_HIP(hipMalloc(&d_chatter, sizeof(uint64_t) * n_streams));
  hipError_t err = hipSuccess;
  if (n_samples > 0)
    err = hipMemcpy2D(d_in, sizeof(float) * n_samples, in, sizeof(float) * stream_stride, sizeof(float) * n_samples, n_streams,
                      hipMemcpyHostToDevice);
  af::GateArgs g{};
  g.in = d_in; g.out = d_in; g.gain_trace = d_trace; g.chatter = d_chatter; g.state = nullptr;
  g.threshold_db = threshold_db;
  g.attack_coeff = af::time_constant_to_coeff(attack_ms, sample_rate);    // gate.rs:160-162
  g.release_coeff = af::time_constant_to_coeff(release_ms, sample_rate);
  g.rms_coeff = af::time_constant_to_coeff(8.0, sample_rate);
  g.n_samples = n_samples; g.stride = n_samples; g.n_streams = n_streams; g.block = trace_block; g.vad_mode = vad_mode ? 1 : 0;
  g.hold_samples = (int32_t)std::llround(sample_rate * 50.0 / 1000.0);
  g.window_samples = (int32_t)std::llround(sample_rate * 500.0 / 1000.0);
  g.cooldown_samples = (int32_t)std::llround(sample_rate * 1000.0 / 1000.0);
  g.relax_samples = (int32_t)std::llround(sample_rate * 700.0 / 1000.0);
  if (err == hipSuccess) err = af::launch_gate(g, nullptr);
  if (err == hipSuccess && n_samples > 0)
    err = hipMemcpy2D(out, sizeof(float) * stream_stride, d_in, sizeof(float) * n_samples, sizeof(float) * n_samples, n_streams,
                      hipMemcpyDeviceToHost);
  if (err == hipSuccess && gain_trace && blocks > 0)
    err = hipMemcpy(gain_trace, d_trace, sizeof(float) * blocks * n_streams, hipMemcpyDeviceToHost);
  if (err == hipSuccess && chatter_events) err = hipMemcpy(chatter_events, d_chatter, sizeof(uint64_t) * n_streams, hipMemcpyDeviceToHost);
  (void)hipFree(d_in);
  (void)hipFree(d_trace);
  (void)hipFree(d_chatter);
  if (err != hipSuccess) return fail(AF_ERR_BACKEND, "gate failed: %s", hipGetErrorString(err));
  return AF_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// NoiseSuppressor (rust-core/src/dsp/noise_suppressor.rs:89-194) for a batch of streams: the trait's ring surface over an
// engine that runs the suppressor alone.  All streams advance in lock step, so the two fixed rings of
// RNNoiseProcessor (rnnoise.rs:11,27-28; audio/rt.rs:146-248) are one pair of [stream][capacity] host buffers with shared
// counters; whole frames go through the GPU in one call per process_frames().
struct af_noise_suppressor {
  af_engine *engine = nullptr;
  int32_t model = AF_NOISE_MODEL_RNNOISE;
  int32_t n_streams = 0;
  bool enabled = true;  // rnnoise.rs:58
  float strength = 1.0f;
  std::vector<float> in_ring, out_ring;  // [stream][kCapacity], linear (compacted on pop)
  int64_t in_len = 0, out_len = 0;
  std::vector<float> scratch_in, scratch_out;
  static constexpr int64_t kCapacity = 8192 + af::kRnnFrame;  // RNNOISE_BUFFER_CAPACITY, rnnoise.rs:11
};

extern "C" {

int af_noise_model_from_id(const char *id, int32_t *model) {  // NoiseModel::from_id, noise_suppressor.rs:58-67
  if (!id || !model) return fail(AF_ERR_INVALID_ARGUMENT, "null argument");
  std::string lower(id);
  for (char &c : lower) c = (char)std::tolower((unsigned char)c);
  if (lower == "rnnoise") { *model = AF_NOISE_MODEL_RNNOISE; return AF_OK; }
  if (lower == "deepfilter-ll" || lower == "deepfilterll") { *model = AF_NOISE_MODEL_DEEPFILTER_LL; return AF_OK; }
  if (lower == "deepfilter" || lower == "deepfilternet") { *model = AF_NOISE_MODEL_DEEPFILTER; return AF_OK; }
  return fail(AF_ERR_INVALID_ARGUMENT, "unknown noise model id '%s'", id);
}
const char *af_noise_model_id(int32_t model) {  // noise_suppressor.rs:47-55
  switch (model) {
    case AF_NOISE_MODEL_RNNOISE: return "rnnoise";
    case AF_NOISE_MODEL_DEEPFILTER_LL: return "deepfilter-ll";
    case AF_NOISE_MODEL_DEEPFILTER: return "deepfilter";
    default: return "";
  }
}
const char *af_noise_model_display_name(int32_t model) {  // noise_suppressor.rs:36-44
  switch (model) {
    case AF_NOISE_MODEL_RNNOISE: return "RNNoise (Low Latency)";
    case AF_NOISE_MODEL_DEEPFILTER_LL: return "DeepFilterNet LL (Fast)";
    case AF_NOISE_MODEL_DEEPFILTER: return "DeepFilterNet (Best Quality)";
    default: return "";
  }
}
int32_t af_noise_model_available(int32_t *models, int32_t capacity) {  // NoiseModel::available, noise_suppressor.rs:70-84
  // the DeepFilterNet variants exist in the reference only behind its `deepfilter` feature and a runtime-loaded df library +
  // model archives; neither is built here, so the list is what a default build of the reference returns
  if (models && capacity > 0) models[0] = AF_NOISE_MODEL_RNNOISE;
  return 1;
}

int af_noise_suppressor_create(int32_t model, int32_t n_streams, int32_t device, af_noise_suppressor **out) {
  if (!out) return fail(AF_ERR_INVALID_ARGUMENT, "out is null");
  *out = nullptr;
  if (model == AF_NOISE_MODEL_DEEPFILTER_LL || model == AF_NOISE_MODEL_DEEPFILTER)
    return fail(AF_ERR_UNSUPPORTED, "the DeepFilterNet backend is not built: its model archives and runtime library are not "
                                    "part of the reference checkout (deepfilter_ffi.rs:9-16); use 'rnnoise'");
  if (model != AF_NOISE_MODEL_RNNOISE) return fail(AF_ERR_INVALID_ARGUMENT, "unknown noise model %d", model);
  af_engine *e = nullptr;
  if (int rc = af_engine_create(48000.0, n_streams, device, &e)) return rc;
  e->proto.eq_enabled = e->proto.eq.enabled = false;
  e->proto.compressor_enabled = e->proto.compressor.enabled = false;
  e->proto.limiter_enabled = e->proto.limiter.enabled = false;
  e->proto.input_scrub = false;  // scale_sample_for_model zeroes non-finite model input itself (rnnoise.rs:90-93)
  e->proto.control_block = af::kRnnFrame;
  e->supp.enabled = true;
  af_noise_suppressor *s = new af_noise_suppressor();
  s->engine = e;
  s->model = model;
  s->n_streams = n_streams;
  s->in_ring.assign((size_t)n_streams * af_noise_suppressor::kCapacity, 0.0f);
  s->out_ring.assign((size_t)n_streams * af_noise_suppressor::kCapacity, 0.0f);
  *out = s;
  return AF_OK;
}
void af_noise_suppressor_destroy(af_noise_suppressor *s) {
  if (!s) return;
  af_engine_destroy(s->engine);
  delete s;
}
af_engine *af_noise_suppressor_engine(af_noise_suppressor *s) { return s ? s->engine : nullptr; }

// push_samples: `samples` is [stream][stride]; returns how many samples per stream the fixed input ring accepted
int64_t af_noise_suppressor_push_samples(af_noise_suppressor *s, const float *samples, int64_t n, int64_t stride) {
  if (!s || (!samples && n > 0) || n < 0 || stride < n) return fail(AF_ERR_INVALID_ARGUMENT, "bad push_samples arguments");
  const int64_t cap = af_noise_suppressor::kCapacity;
  const int64_t written = std::min<int64_t>(n, cap - s->in_len);  // FixedAudioRing::push_slice, rt.rs:189-197
  for (int32_t k = 0; k < s->n_streams; ++k)
    std::memcpy(&s->in_ring[(size_t)k * cap + s->in_len], samples + (size_t)k * stride, sizeof(float) * written);
  s->in_len += written;
  return written;
}

static void ring_consume(std::vector<float> &ring, int64_t &len, int64_t count, int32_t n_streams) {
  const int64_t cap = af_noise_suppressor::kCapacity;
  if (count <= 0) return;
  for (int32_t k = 0; k < n_streams; ++k)
    std::memmove(&ring[(size_t)k * cap], &ring[(size_t)k * cap + count], sizeof(float) * (len - count));
  len -= count;
}

int af_noise_suppressor_process_frames(af_noise_suppressor *s) {  // rnnoise.rs:122-164
  if (!s) return fail(AF_ERR_INVALID_ARGUMENT, "suppressor is null");
  const int64_t cap = af_noise_suppressor::kCapacity;
  if (!s->enabled) {  // bypass: input_buffer.move_into(&mut output_buffer), rnnoise.rs:123-126
    const int64_t moved = std::min<int64_t>(s->in_len, cap - s->out_len);
    for (int32_t k = 0; k < s->n_streams; ++k)
      std::memcpy(&s->out_ring[(size_t)k * cap + s->out_len], &s->in_ring[(size_t)k * cap], sizeof(float) * moved);
    s->out_len += moved;
    ring_consume(s->in_ring, s->in_len, moved, s->n_streams);
    return AF_OK;
  }
  // while input.len() >= 480 && output.remaining() >= 480
  const int64_t frames = std::min<int64_t>(s->in_len / af::kRnnFrame, (cap - s->out_len) / af::kRnnFrame);
  if (frames <= 0) return AF_OK;
  const int64_t n = frames * af::kRnnFrame;
  s->scratch_in.resize((size_t)s->n_streams * n);
  s->scratch_out.resize((size_t)s->n_streams * n);
  for (int32_t k = 0; k < s->n_streams; ++k)
    std::memcpy(&s->scratch_in[(size_t)k * n], &s->in_ring[(size_t)k * cap], sizeof(float) * n);
  if (int rc = af_engine_set_suppressor_strength(s->engine, s->strength)) return rc;
  if (int rc = af_engine_process_host(s->engine, s->scratch_in.data(), s->scratch_out.data(), n, AF_LAYOUT_STREAM_MAJOR)) return rc;
  for (int32_t k = 0; k < s->n_streams; ++k)
    std::memcpy(&s->out_ring[(size_t)k * cap + s->out_len], &s->scratch_out[(size_t)k * n], sizeof(float) * n);
  s->out_len += n;
  ring_consume(s->in_ring, s->in_len, n, s->n_streams);
  return AF_OK;
}

int64_t af_noise_suppressor_available_samples(const af_noise_suppressor *s) { return s ? s->out_len : 0; }
int64_t af_noise_suppressor_pending_input(const af_noise_suppressor *s) { return s ? s->in_len : 0; }

// pop_samples_into / read_samples (rnnoise.rs:185-188): up to `count` samples per stream into out[stream][stride]
int64_t af_noise_suppressor_pop_samples_into(af_noise_suppressor *s, float *out, int64_t count, int64_t stride) {
  if (!s || (!out && count > 0) || count < 0 || stride < count) return fail(AF_ERR_INVALID_ARGUMENT, "bad pop_samples_into arguments");
  const int64_t cap = af_noise_suppressor::kCapacity;
  const int64_t n = std::min<int64_t>(count, s->out_len);
  for (int32_t k = 0; k < s->n_streams; ++k) std::memcpy(out + (size_t)k * stride, &s->out_ring[(size_t)k * cap], sizeof(float) * n);
  ring_consume(s->out_ring, s->out_len, n, s->n_streams);
  return n;
}
int64_t af_noise_suppressor_drain_pending_input(af_noise_suppressor *s, float *out, int64_t capacity, int64_t stride) {  // rnnoise.rs:240-244
  if (!s || (!out && capacity > 0) || capacity < 0 || stride < capacity) return fail(AF_ERR_INVALID_ARGUMENT, "bad drain_pending_input arguments");
  const int64_t cap = af_noise_suppressor::kCapacity;
  const int64_t n = std::min<int64_t>(capacity, s->in_len);
  for (int32_t k = 0; k < s->n_streams; ++k) std::memcpy(out + (size_t)k * stride, &s->in_ring[(size_t)k * cap], sizeof(float) * n);
  ring_consume(s->in_ring, s->in_len, n, s->n_streams);
  return n;
}

int af_noise_suppressor_set_strength(af_noise_suppressor *s, float value) {  // rnnoise.rs:67-72
  if (!s) return fail(AF_ERR_INVALID_ARGUMENT, "suppressor is null");
  s->strength = af::clampf(value, 0.0f, 1.0f);
  return AF_OK;
}
float af_noise_suppressor_get_strength(const af_noise_suppressor *s) { return s ? s->strength : 0.0f; }
int af_noise_suppressor_set_enabled(af_noise_suppressor *s, int32_t enabled) {  // rnnoise.rs:194-196: state is kept
  if (!s) return fail(AF_ERR_INVALID_ARGUMENT, "suppressor is null");
  s->enabled = enabled != 0;
  return AF_OK;
}
int32_t af_noise_suppressor_is_enabled(const af_noise_suppressor *s) { return s && s->enabled ? 1 : 0; }
int af_noise_suppressor_soft_reset(af_noise_suppressor *s) {  // flush_buffers, rnnoise.rs:216-232: model state survives
  if (!s) return fail(AF_ERR_INVALID_ARGUMENT, "suppressor is null");
  s->in_len = s->out_len = 0;
  return AF_OK;
}
int af_noise_suppressor_reset(af_noise_suppressor *s) {  // rnnoise.rs:205-210: a new DenoiseState + empty rings
  if (!s) return fail(AF_ERR_INVALID_ARGUMENT, "suppressor is null");
  s->in_len = s->out_len = 0;
  if (!s->engine->started) return AF_OK;
  AF_HIP(hipSetDevice(s->engine->device));
  AF_HIP(hipDeviceSynchronize());
  // the wet/dry smoothing state belongs to the wrapper, not to DenoiseState: it survives (rnnoise.rs:205-210)
  std::vector<float> smoothed((size_t)s->n_streams);
  AF_HIP(hipMemcpy2D(smoothed.data(), sizeof(float), s->engine->supp.d_state + af::SuppState::kSmoothedStrength,
                     sizeof(float) * af::SuppState::kCount, sizeof(float), s->n_streams, hipMemcpyDeviceToHost));
  AF_HIP(s->engine->supp.reset_state(s->n_streams));
  AF_HIP(hipMemcpy2D(s->engine->supp.d_state + af::SuppState::kSmoothedStrength, sizeof(float) * af::SuppState::kCount, smoothed.data(),
                     sizeof(float), sizeof(float), s->n_streams, hipMemcpyHostToDevice));
  return AF_OK;
}
int32_t af_noise_suppressor_model_type(const af_noise_suppressor *s) { return s ? s->model : -1; }
int32_t af_noise_suppressor_latency_samples(const af_noise_suppressor *) { return af::kRnnFrame; }  // rnnoise.rs:313-315
int32_t af_noise_suppressor_backend_available(const af_noise_suppressor *s) { return s ? 1 : 0; }   // rnnoise.rs:317-319
int32_t af_noise_suppressor_backend_failed(const af_noise_suppressor *) { return 0; }               // rnnoise.rs:325-327
const char *af_noise_suppressor_backend_error(const af_noise_suppressor *) { return nullptr; }      // rnnoise.rs:321-323

}  // extern "C"
