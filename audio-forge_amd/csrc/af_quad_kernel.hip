// af_quad_kernel.hip -- Kernel 3: chain_quad_kernel ("token ring, four samples across a quad")
//
// The token ring of kernel 2 (af_ring_kernel.hip) keeps every stream strictly sequential through every
// recurrence but gives one workgroup 64 streams, so batch 4096 occupies 64 of the 256 CUs -- and profiling
// the stages one by one showed where a CU's time goes: ~85 % of the issue slots are the FEED-FORWARD
// double-precision math between the compressor's recurrences (four log10, three exp10, four sqrt, two
// divisions per sample) and the three 128-tap true-peak FIRs, none of which carries state.
//
// This kernel keeps the ring (16 waves, one 4-sample chunk each, a token per recurrence) and re-cuts the lanes:
// a workgroup owns 16 streams, lane = 4 * stream + k, and the four lanes of a quad hold the four samples of
// the wave's chunk.
//   * feed-forward zones run once per lane on the lane's own sample: a quarter of the instructions of
//     kernel 2 for the same work, all 64 lanes busy;
//   * recurrences need the chunk in order: the quad's four values are exchanged with DPP quad_perm
//     broadcasts (register-to-register, no LDS), after which every lane of the quad walks the same four
//     steps on the same inputs and all four hold the same state -- the recurrences are cheap, the
//     redundancy costs less than masking would;
//   * per-stream state sits in LDS as [row][16 streams]; 37 KB per workgroup instead of 150 KB.
// Batch 4096 is now 256 workgroups -- one per CU -- and a CU spends ~2.5x fewer issue slots per
// stream-sample.  Arithmetic, operation order and rounding points per stream are exactly those of kernels
// 1 and 2 (and of the reference); the same parity tests run against all three.
//
// Not built into this kernel: the compressor's auto-makeup controller and the two-launch pre-pass mode; the
// host routes those configurations to kernel 2.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "af_dsp.h"

namespace af {

namespace quad {

constexpr int kQ = 4;               // samples of a chunk = lanes of a quad
constexpr int kStreams = kLanes / kQ;  // streams of a workgroup
constexpr int kTpRing = 128;
constexpr int kEqGroup = 5;
constexpr int kMaxEqGroups = kMaxEqSections / kEqGroup;

enum : int { kTokIn = 0, kTokCompA, kTokCompC, kTokCompE, kTokLim, kTokTp, kTokFin, kTokEq0, kNumTokens = kTokEq0 + kMaxEqGroups };

// LDS rows, f64 plane (each row = 16 doubles)
enum : int {
  kR64PreZ1 = 0, kR64PreZ2,
  kR64ScPrevIn, kR64ScPrevOut, kR64LowEnv, kR64VoicedEnv, kR64PresenceEnv, kR64Plosive,
  kR64PeakEnvDb, kR64RmsEnvSq, kR64Gr, kR64FastEnv, kR64SlowEnv, kR64CurReleaseMs, kR64TargetReleaseMs,
  kR64ReleaseCoeff, kR64SmoothedMakeup, kR64MakeupLin,
  kR64LimGain, kR64LimGmin, kR64InSq, kR64OutSq,
  kR64Eq  // then 2 rows per section (z1 z2), followed by 2 more per section (pz1 pz2) while a crossfade is pending
};
// LDS rows, f32 plane (each row = 16 floats)
enum : int {
  kR32DcX1 = 0, kR32DcY1, kR32TpGain, kR32LimPrefix, kR32InPeak, kR32OutPeak, kR32TpInPeak, kR32OutTp,
  kR32TpGmin, kR32TpLimited, kR32NonFinite, kR32Pad,
  kR32Tpi,                     // kTpRing rows
  kR32Tpo = kR32Tpi + kTpRing, // kTpRing rows
  kR32LimRing = kR32Tpo + kTpRing  // 2*W rows
};

__host__ __device__ inline size_t lds_bytes(int n_sections, int lookahead, bool crossfade) {
  const size_t rows64 = kR64Eq + (crossfade ? 4 : 2) * (size_t)n_sections;
  const size_t rows32 = kR32LimRing + 2 * ((size_t)lookahead + 1);
  return 256 + rows64 * kStreams * sizeof(double) + rows32 * kStreams * sizeof(float);
}

constexpr int kAbortSlot = 32;
__device__ __forceinline__ void token_wait(int *turn_base, int tok, int q) {
  int spins = 0;
  while (__hip_atomic_load(&turn_base[tok], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != q) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 0xfff) == 0) {
      if (__hip_atomic_load(&turn_base[kAbortSlot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return;
      if (spins > (1 << 25)) {
        __hip_atomic_store(&turn_base[kAbortSlot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
      }
    }
  }
  // (raising the wave's priority inside a serial unit, which gains 7 % in the ring kernel, costs 2 % here: with four
  // samples across a quad the serial units are a larger share of every wave's work)
}
__device__ __forceinline__ void token_pass(int *turn_base, int tok, int q) {
  __hip_atomic_store(&turn_base[tok], q + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- quad exchange: every lane of a quad gets the value lane J of the quad holds (DPP quad_perm [J,J,J,J])
template <int J>
__device__ __forceinline__ float qb(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, false));
}
template <int J>
__device__ __forceinline__ double qb(double v) {
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffll), J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <class T>
__device__ __forceinline__ void gather(T own, T (&arr)[kQ]) {
  arr[0] = qb<0>(own);
  arr[1] = qb<1>(own);
  arr[2] = qb<2>(own);
  arr[3] = qb<3>(own);
}
template <class T>
__device__ __forceinline__ T pick(const T (&arr)[kQ], int k) {
  return k == 0 ? arr[0] : (k == 1 ? arr[1] : (k == 2 ? arr[2] : arr[3]));
}

// Bandlimited4xPeak::observe (true_peak.rs:173-186) over a shared ring: sample n sits in row n & 127
__device__ __forceinline__ float tp_observe_ring(const float *ring, int n, int sl) {
  float h[kTpTaps];
#pragma unroll
  for (int k = 0; k < kTpTaps; ++k) h[k] = ring[((n - k) & (kTpRing - 1)) * kStreams + sl];
  float peak = fabsf(h[0]);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < kTpTaps; ++k) acc = __builtin_fmaf(AF_TP_FIR[p][k], h[k], acc);
    peak = fmaxf(peak, fabsf(acc));
  }
  return peak;
}

}  // namespace quad

template <int kRingWaves>
__global__ __launch_bounds__(kRingWaves *kLanes) void chain_quad_kernel(LaunchArgs a) {
  using namespace quad;
  constexpr int kChunk = kQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const ChainParams &P = *a.params;
  const uint32_t flags = P.flags;
  const int nsec = (flags & kFlagEq) ? P.n_eq_sections : 0;
  const int n_groups = (nsec + kEqGroup - 1) / kEqGroup;
  const int W = P.lim.lookahead_samples + 1;

  int *turn = reinterpret_cast<int *>(lds_raw);
  double *l64 = reinterpret_cast<double *>(lds_raw + 256);
  bool any_xf = false;
  for (int k = 0; k < P.n_eq_sections; ++k) any_xf |= P.eq[k].xf_remaining > 0;
  const int rows64 = kR64Eq + (any_xf ? 4 : 2) * P.n_eq_sections;
  const int pz_base = kR64Eq + 2 * P.n_eq_sections;  // rows of the pending-filter memories
  float *l32 = reinterpret_cast<float *>(lds_raw + 256 + (size_t)rows64 * kStreams * sizeof(double));
#define L64(row) l64[(row)*kStreams + sl]
#define L32(row) l32[(row)*kStreams + sl]

  const int tid = threadIdx.x;
  const int lane = tid & (kLanes - 1);
  const int sl = lane >> 2;  // stream of this lane inside the workgroup
  const int kq = lane & 3;   // which sample of the chunk this lane owns in the feed-forward zones
  const int wave = tid / kLanes;
  const int s0 = blockIdx.x * kStreams;
  const int s = s0 + sl;
  const bool valid = s < a.n_streams;
  const bool writer = valid && kq == 0;  // one lane of the quad writes the stream's results to HBM
  const int sc = valid ? s : a.n_streams - 1;
  const int64_t NS = a.n_streams;
  const int64_t n0 = a.samples_before;  // absolute index of this launch's first sample

  // ---------------- stage the per-stream state into LDS (wave w takes rows w, w+16, ...)
  if (tid < 64) turn[tid] = 0;
  {
    struct Map { int row, field; };
    const Map m64[] = {{kR64PreZ1, kPreZ1}, {kR64PreZ2, kPreZ2}, {kR64ScPrevIn, kCompScPrevIn},
                       {kR64ScPrevOut, kCompScPrevOut}, {kR64LowEnv, kCompLowEnv}, {kR64VoicedEnv, kCompVoicedEnv},
                       {kR64PresenceEnv, kCompPresenceEnv}, {kR64Plosive, kCompPlosive}, {kR64PeakEnvDb, kCompPeakEnvDb},
                       {kR64RmsEnvSq, kCompRmsEnvSq}, {kR64Gr, kCompGr}, {kR64FastEnv, kCompFastEnv},
                       {kR64SlowEnv, kCompSlowEnv}, {kR64CurReleaseMs, kCompCurReleaseMs},
                       {kR64TargetReleaseMs, kCompTargetReleaseMs}, {kR64ReleaseCoeff, kCompReleaseCoeff},
                       {kR64SmoothedMakeup, kCompSmoothedMakeup}, {kR64LimGain, kLimGain}};
    const int n_m64 = (int)(sizeof(m64) / sizeof(m64[0]));
    for (int k = wave; k < n_m64; k += kRingWaves) L64(m64[k].row) = a.st64[(int64_t)m64[k].field * NS + sc];
    for (int k = wave; k < (any_xf ? 4 : 2) * nsec; k += kRingWaves) {
      const int sec = k >> (any_xf ? 2 : 1), part = k & (any_xf ? 3 : 1);  // state plane: z1 z2 pz1 pz2 per section
      const int row = part < 2 ? kR64Eq + 2 * sec + part : pz_base + 2 * sec + (part - 2);
      L64(row) = a.st64[(int64_t)(kEqBase + 4 * sec + part) * NS + sc];
    }
    if (wave == 0) {
      L64(kR64MakeupLin) = db2lin(a.st64[(int64_t)kCompSmoothedMakeup * NS + sc]);
      L64(kR64LimGmin) = 1.0;
      L64(kR64InSq) = 0.0;
      L64(kR64OutSq) = 0.0;
      L32(kR32DcX1) = a.st32[(int64_t)kDcX1 * NS + sc];
      L32(kR32DcY1) = a.st32[(int64_t)kDcY1 * NS + sc];
      L32(kR32TpGain) = a.st32[(int64_t)kTpGain * NS + sc];
      L32(kR32LimPrefix) = a.st32[(int64_t)kLimPrefix * NS + sc];
      L32(kR32InPeak) = 0.0f;
      L32(kR32OutPeak) = 0.0f;
      L32(kR32TpInPeak) = 0.0f;
      L32(kR32OutTp) = 0.0f;
      L32(kR32TpGmin) = 1.0f;
      L32(kR32TpLimited) = 0.0f;
      L32(kR32NonFinite) = 0.0f;
    }
    // history rows: state row r holds sample n0-32+r
    for (int r = wave; r < kTpTaps; r += kRingWaves) {
      const int row = (int)((n0 - kTpTaps + r) & (kTpRing - 1));
      L32(kR32Tpi + row) = a.st32[(int64_t)(kTpInHist + r) * NS + sc];
      L32(kR32Tpo + row) = a.st32[(int64_t)(kTpOutHist + r) * NS + sc];
    }
    if (flags & kFlagLimiter)
      for (int r = wave; r < 2 * W; r += kRingWaves) L32(kR32LimRing + r) = a.st32[(int64_t)(kLimRing + r) * NS + sc];
  }
  __syncthreads();

  const int cb = P.control_block;
  const int cpb = (cb + kChunk - 1) / kChunk;  // chunks per full control block
  const int64_t n_blocks = (a.n_samples + cb - 1) / cb;
  const int64_t last_len = a.n_samples - (n_blocks - 1) * cb;
  const int64_t Q = n_blocks > 0 ? (n_blocks - 1) * cpb + (last_len + kChunk - 1) / kChunk : 0;
  const float tp_ceiling = P.tp.ceiling_linear;

  for (int64_t q64 = wave; q64 < Q; q64 += kRingWaves) {
    const int q = (int)q64;
    const int64_t b = q64 / cpb;
    const int i = (int)(q64 - b * cpb);
    const int blk_len = (int)((a.n_samples - b * cb) < cb ? (a.n_samples - b * cb) : cb);
    const int64_t t0 = b * cb + (int64_t)i * kChunk;  // launch-relative index of the chunk's first sample
    const int len = (blk_len - i * kChunk) < kChunk ? (blk_len - i * kChunk) : kChunk;
    const bool first_in_block = i == 0;
    const bool last_in_block = i * kChunk + len == blk_len;
    const bool own = kq < len;  // this lane's sample exists
    BlockStats *row = a.stats ? &a.stats[b * NS + sc] : nullptr;
    const int nb = (int)((n0 + t0) & (kTpRing - 1));  // ring row of the chunk's first sample

    // ---- load the lane's sample (issued before the first token wait); scrub / clamp are pointwise
    float xo = 0.0f;
    if (valid && own) xo = a.layout == 0 ? a.in[(int64_t)s * a.stream_stride + t0 + kq] : a.in[(t0 + kq) * a.stream_stride + s];
    if ((flags & (kFlagInputScrub | kFlagInputClamp)) && !finite_f32(xo)) xo = 0.0f;
    if (flags & kFlagInputClamp) xo = fclamp(xo, -1.0f, 1.0f);
    float x[kChunk];
    gather(xo, x);

    // =========================== token: block input stats, DC block + fixed HP
    token_wait(turn, kTokIn, q);
    {
      double in_sq = first_in_block ? 0.0 : L64(kR64InSq);
      float in_peak = first_in_block ? 0.0f : L32(kR32InPeak);
#pragma unroll
      for (int k = 0; k < kChunk; ++k)
        if (k < len) {
          const float v = x[k];
          in_sq += (double)v * (double)v;
          in_peak = fmaxf(in_peak, fabsf(v));
        }
      L64(kR64InSq) = in_sq;
      L32(kR32InPeak) = in_peak;
      if (last_in_block && writer && row) {
        row->input_square_sum = in_sq;
        row->input_sample_peak = in_peak;
      }
      if (flags & kFlagDcBlock) {  // routing.rs:826-843
        float dc_x1 = L32(kR32DcX1), dc_y1 = L32(kR32DcY1);
        double z1 = L64(kR64PreZ1), z2 = L64(kR64PreZ2);
        const BiquadCoef c = P.pre_hp;
#pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (k < len) {
            const float in = x[k];
            const float o = in - dc_x1 + 0.995f * dc_y1;
            dc_x1 = in;
            dc_y1 = o;
            float r = o;
            if (flags & kFlagPreHighpass) {
              const double xin = (double)o;
              const double y = c.b0 * xin + z1;
              z1 = c.b1 * xin - c.a1 * y + z2;
              z2 = c.b2 * xin - c.a2 * y;
              r = (float)y;
            }
            x[k] = r;
          }
        L32(kR32DcX1) = dc_x1;
        L32(kR32DcY1) = dc_y1;
        L64(kR64PreZ1) = z1;
        L64(kR64PreZ2) = z2;
      }
    }
    token_pass(turn, kTokIn, q);

    // =========================== tokens: EQ section groups (eq.rs:371-379, biquad.rs:263-327)
    for (int g = 0; g < n_groups; ++g) {
      const int k0 = g * kEqGroup;
      const int k1 = (k0 + kEqGroup) < nsec ? (k0 + kEqGroup) : nsec;
      token_wait(turn, kTokEq0 + g, q);
      for (int ks = k0; ks < k1; ++ks) {
        const SectionParams &sp = P.eq[ks];
        double z1 = L64(kR64Eq + 2 * ks), z2 = L64(kR64Eq + 2 * ks + 1);
        BiquadCoef c = sp.active;
        int rem = sp.xf_remaining - (int)(t0 < sp.xf_remaining ? t0 : sp.xf_remaining);
        if (rem > 0) {
          const BiquadCoef p = sp.pending;
          double pz1 = L64(pz_base + 2 * ks), pz2 = L64(pz_base + 2 * ks + 1);
          const double total = (double)sp.xf_total;
#pragma unroll
          for (int k = 0; k < kChunk; ++k)
            if (k < len) {
              const double in = (double)x[k];
              const double ya = c.b0 * in + z1;
              z1 = c.b1 * in - c.a1 * ya + z2;
              z2 = c.b2 * in - c.a2 * ya;
              double y = ya;
              if (rem > 0) {
                const double yp = p.b0 * in + pz1;
                pz1 = p.b1 * in - p.a1 * yp + pz2;
                pz2 = p.b2 * in - p.a2 * yp;
                const double fade = (double)(sp.xf_total - rem + 1) / total;
                y = ya * (1.0 - fade) + yp * fade;
                rem -= 1;
                if (rem == 0) {
                  c = p;
                  z1 = pz1;
                  z2 = pz2;
                }
              }
              x[k] = (float)y;
            }
          L64(pz_base + 2 * ks) = pz1;
          L64(pz_base + 2 * ks + 1) = pz2;
        } else {
          if (sp.xf_remaining > 0) c = sp.pending;
#pragma unroll
          for (int k = 0; k < kChunk; ++k)
            if (k < len) {
              const double in = (double)x[k];
              const double y = c.b0 * in + z1;
              z1 = c.b1 * in - c.a1 * y + z2;
              z2 = c.b2 * in - c.a2 * y;
              x[k] = (float)y;
            }
        }
        L64(kR64Eq + 2 * ks) = z1;
        L64(kR64Eq + 2 * ks + 1) = z2;
      }
      token_pass(turn, kTokEq0 + g, q);
    }

    // =========================== compressor (compressor.rs:700-774)
    if (flags & kFlagCompressor) {
      const CompressorParams &cp = P.comp;
      double d[kChunk], low_e[kChunk], voiced_e[kChunk], presence_e[kChunk], rms_e[kChunk];
      // ---- token A: side-chain high-pass + band / rms envelopes (linear recurrences)
      token_wait(turn, kTokCompA, q);
      {
        double rms_env = L64(kR64RmsEnvSq);
        if (cp.sidechain_highpass_enabled) {
          double prev_in = L64(kR64ScPrevIn), prev_out = L64(kR64ScPrevOut);
          double low_env = L64(kR64LowEnv), voiced_env = L64(kR64VoicedEnv), presence_env = L64(kR64PresenceEnv);
          const double kk = cp.band_env_coeff;
#pragma unroll
          for (int k = 0; k < kChunk; ++k)
            if (k < len) {
              const double xin = (double)x[k];
              const double dd = cp.sidechain_highpass_coeff * (prev_out + xin - prev_in);
              prev_in = xin;
              prev_out = dd;
              const double low = xin - dd;
              const double presence = 0.65 * dd + 0.35 * (dd - low);
              low_env = kk * low_env + (1.0 - kk) * low * low;
              voiced_env = kk * voiced_env + (1.0 - kk) * dd * dd;
              presence_env = kk * presence_env + (1.0 - kk) * presence * presence;
              rms_env = cp.rms_coeff * rms_env + (1.0 - cp.rms_coeff) * (dd * dd);
              d[k] = dd;
              low_e[k] = low_env;
              voiced_e[k] = voiced_env;
              presence_e[k] = presence_env;
              rms_e[k] = rms_env;
            }
          L64(kR64ScPrevIn) = prev_in;
          L64(kR64ScPrevOut) = prev_out;
          L64(kR64LowEnv) = low_env;
          L64(kR64VoicedEnv) = voiced_env;
          L64(kR64PresenceEnv) = presence_env;
        } else {
#pragma unroll
          for (int k = 0; k < kChunk; ++k)
            if (k < len) {
              const double dd = (double)x[k];
              rms_env = cp.rms_coeff * rms_env + (1.0 - cp.rms_coeff) * (dd * dd);
              d[k] = dd;
              rms_e[k] = rms_env;
            }
        }
        L64(kR64RmsEnvSq) = rms_env;
      }
      token_pass(turn, kTokCompA, q);
      // ---- feed-forward, own sample: detector weight, instantaneous peak and RMS levels in dB
      double weight_own = kDetectorUnitWeight, plosive_own = 0.0;
      if (cp.sidechain_highpass_enabled) {  // update_sidechain_band_metrics, compressor.rs:438-449
        const double low_rms = sqrt(pick(low_e, kq));
        const double voiced_rms = fmax(sqrt(pick(voiced_e, kq)), 1e-8);
        const double presence_rms = sqrt(pick(presence_e, kq));
        const double plosive = dclamp(low_rms / voiced_rms, 0.0, 32.0);
        plosive_own = plosive;
        const double plosive_amount = dclamp(div_known(plosive - 1.25, 3.75, 1.0 / 3.75), 0.0, 1.0);
        const double plosive_penalty = 1.0 - plosive_amount * (1.0 - 0.35);
        const double presence_ratio = dclamp(presence_rms / voiced_rms, 0.0, 4.0);
        const double presence_weight = 1.0 + 0.18 * dclamp(presence_ratio - 0.75, 0.0, 1.0);
        weight_own = detector_weight(dclamp(plosive_penalty * presence_weight, 0.35, 1.15));
      }
      const double inst_peak_own = lin2db(fabs(pick(d, kq)), 1e-10);
      const double rms_db_own = detector_rms_level(pick(rms_e, kq));
      double inst_peak_db[kChunk], plosive_k[kChunk];
      gather(inst_peak_own, inst_peak_db);
      gather(plosive_own, plosive_k);
      // ---- token C: log-domain peak envelope (compressor.rs:735-742)
      double peak_db[kChunk];
      token_wait(turn, kTokCompC, q);
      {
        double pe = L64(kR64PeakEnvDb);
#pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (k < len) {
            const double pk = inst_peak_db[k] > pe ? cp.attack_coeff : cp.detector_release_coeff;
            pe = pk * pe + (1.0 - pk) * inst_peak_db[k];
            peak_db[k] = pe;
          }
        L64(kR64PeakEnvDb) = pe;
        if (len > 0) L64(kR64Plosive) = cp.sidechain_highpass_enabled ? pick(plosive_k, len - 1) : 0.0;  // diagnostic state only
      }
      token_pass(turn, kTokCompC, q);
      // ---- feed-forward, own sample: blended detector level -> static gain-reduction target
      double target[kChunk];
      {
        gather(comp_gain_reduction(cp, detector_db(pick(peak_db, kq), rms_db_own, weight_own)), target);
      }
      // ---- token E: release-time meter + gain-reduction smoothing (compressor.rs:452-505,752-764)
      double gr_k[kChunk];
      double makeup_lin;
      token_wait(turn, kTokCompE, q);
      {
        double gr = L64(kR64Gr), fast = L64(kR64FastEnv), slow = L64(kR64SlowEnv);
        double cur_ms = L64(kR64CurReleaseMs), tgt_ms = L64(kR64TargetReleaseMs);
        const double rel_coeff = L64(kR64ReleaseCoeff);
        makeup_lin = L64(kR64MakeupLin);
#pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (k < len) {
            if (cp.adaptive_release) {
              const double sustained = dclamp(div_known(slow, 6.0, 1.0 / 6.0), 0.0, 1.0);
              const double transient_bias = dclamp(div_known(fast - slow, 7.0, 1.0 / 7.0), 0.0, 1.0);
              const double syllabic = dclamp(sustained * sustained * (1.0 - 0.35 * transient_bias), 0.0, 1.0);
              tgt_ms = 50.0 + syllabic * (400.0 - 50.0);
            } else {
              tgt_ms = cp.base_release_ms;
            }
            if (fabs(tgt_ms - cur_ms) > 1.0) {
              cur_ms = cp.release_smoothing_coeff * cur_ms + (1.0 - cp.release_smoothing_coeff) * tgt_ms;
            } else {
              cur_ms = tgt_ms;
            }
            const double tg = target[k];
            if (!cp.adaptive_release) {
              const double kk = tg > gr ? cp.attack_coeff : rel_coeff;
              gr = kk * gr + (1.0 - kk) * tg;
              fast = gr;
              slow = 0.0;
            } else {
              if (tg > gr) {
                fast = cp.attack_coeff * gr + (1.0 - cp.attack_coeff) * tg;
              } else {
                fast = cp.fast_release_coeff * fast + (1.0 - cp.fast_release_coeff) * tg;
              }
              if (tg > 3.0) {
                slow = cp.slow_charge_coeff * slow + (1.0 - cp.slow_charge_coeff) * tg;
              } else {
                slow *= cp.slow_release_coeff;
              }
              gr = fmax(fast, slow);
            }
            gr_k[k] = gr;
          }
        L64(kR64Gr) = gr;
        L64(kR64FastEnv) = fast;
        L64(kR64SlowEnv) = slow;
        L64(kR64CurReleaseMs) = cur_ms;
        L64(kR64TargetReleaseMs) = tgt_ms;
        if (last_in_block && writer && row) row->compressor_gr_db = (float)gr;
        if (last_in_block) {  // update_auto_makeup_gain, auto-makeup off (compressor.rs:604-617)
          double sm = L64(kR64SmoothedMakeup);
          const double makeup_coeff = pow(cp.makeup_smoothing_coeff, (double)(blk_len < 1 ? 1 : blk_len));
          const double tgt = cp.makeup_gain_db;
          if (fabs(tgt - sm) > 0.1) {
            sm = makeup_coeff * sm + (1.0 - makeup_coeff) * tgt;
          } else {
            sm = tgt;
          }
          L64(kR64SmoothedMakeup) = sm;
          L64(kR64MakeupLin) = db2lin(sm);
          if (writer && row) row->makeup_gain_db = (float)sm;
        }
      }
      token_pass(turn, kTokCompE, q);
      // ---- feed-forward, own sample: apply gain (compressor.rs:771-773)
      gather((float)((double)pick(x, kq) * (db2lin(-pick(gr_k, kq)) * makeup_lin)), x);
    }

    // =========================== limiter + true-peak limiter
    float itp[kChunk];
    if (flags & kFlagLimiter) {
      const double ceil_lin = P.lim.ceiling_linear;
      const double rc = P.lim.release_coeff;
      float *ring = &l32[kR32LimRing * kStreams];
      float *suf = &l32[(kR32LimRing + W) * kStreams];
      // ---- token: lookahead limiter (limiter.rs:246-284), sliding max by block prefix/suffix maxima
      token_wait(turn, kTokLim, q);
      {
        double g = L64(kR64LimGain);
        double gmin = first_in_block ? 1.0 : L64(kR64LimGmin);
        float prefix = L32(kR32LimPrefix);
        int j = (int)((n0 + t0) % W);
#pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (k < len) {
            const float xin = x[k];
            const float ax = fabsf(xin);
            const int jn = (j + 1 == W) ? 0 : j + 1;
            const float delayed = ring[jn * kStreams + sl];
            const float sfx = (j + 1 < W) ? suf[(j + 1) * kStreams + sl] : 0.0f;
            prefix = (j == 0) ? ax : fmaxf(prefix, ax);
            const double peak = (double)fmaxf(sfx, prefix);
            ring[j * kStreams + sl] = xin;
            if (j + 1 == W) {
              float m = 0.0f;
              for (int kk = W - 1; kk >= 0; --kk) {
                m = fmaxf(m, fabsf(ring[kk * kStreams + sl]));
                suf[kk * kStreams + sl] = m;
              }
            }
            j = jn;
            const double tg = peak > ceil_lin ? ceil_lin / peak : 1.0;
            if (tg < g) {
              g = tg;
            } else {
              g = rc * g + (1.0 - rc) * tg;
            }
            gmin = fmin(gmin, g);
            const float o = (float)dclamp((double)delayed * g, -ceil_lin, ceil_lin);
            x[k] = finite_f32(o) ? o : 0.0f;  // TruePeakLimiter input scrub, true_peak.rs:342
            L32(kR32Tpi + ((nb + k) & (kTpRing - 1))) = x[k];
          }
        L64(kR64LimGain) = g;
        L64(kR64LimGmin) = gmin;
        L32(kR32LimPrefix) = prefix;
        if (last_in_block && writer && row)
          row->limiter_peak_gr_db = gmin < 1.0 ? (float)(-lin2db(gmin, 1e-10)) : 0.0f;
      }
      token_pass(turn, kTokLim, q);
      // ---- feed-forward, own sample: input-side 4x true peak
      gather(own ? tp_observe_ring(&l32[kR32Tpi * kStreams], nb + kq, sl) : 0.0f, itp);
    }

    // ---- token: true-peak gain (true_peak.rs:341-374), chain output, block output stats
    token_wait(turn, kTokTp, q);
    {
      double out_sq = first_in_block ? 0.0 : L64(kR64OutSq);
      float out_peak = first_in_block ? 0.0f : L32(kR32OutPeak);
      float nonfinite = first_in_block ? 0.0f : L32(kR32NonFinite);
      float tp_in_peak = 0.0f, tp_gmin = 1.0f, tp_limited = 0.0f;
      if (flags & kFlagLimiter) {
        float g = L32(kR32TpGain);
        tp_in_peak = first_in_block ? 0.0f : L32(kR32TpInPeak);
        tp_gmin = first_in_block ? 1.0f : L32(kR32TpGmin);
        tp_limited = first_in_block ? 0.0f : L32(kR32TpLimited);
        const float rel = P.tp.release_coeff;
#pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (k < len) {
            const float delayed = L32(kR32Tpi + ((nb + k - kTpDelay) & (kTpRing - 1)));
            tp_in_peak = fmaxf(tp_in_peak, itp[k]);
            float tg = 1.0f;
            if (itp[k] > tp_ceiling) tg = fclamp((tp_ceiling * 0.999f) / itp[k], 0.0f, 1.0f);
            if (tg < g) {
              g = tg;
              tp_limited = 1.0f;
            } else {
              g = rel * g + (1.0f - rel) * tg;
            }
            tp_gmin = fminf(tp_gmin, g);
            float o = fclamp(delayed * g, -tp_ceiling, tp_ceiling);
            if (!finite_f32(o)) o = 0.0f;
            x[k] = o;
          }
        L32(kR32TpGain) = g;
        L32(kR32TpInPeak) = tp_in_peak;
        L32(kR32TpGmin) = tp_gmin;
        L32(kR32TpLimited) = tp_limited;
      }
#pragma unroll
      for (int k = 0; k < kChunk; ++k)
        if (k < len) {
          const float o = x[k];
          float det = o;
          if (finite_f32(o)) {
            out_sq += (double)o * (double)o;
          } else {
            nonfinite = 1.0f;
            det = 0.0f;  // TruePeakDetector::process_block, true_peak.rs:212
          }
          out_peak = fmaxf(out_peak, fabsf(o));
          L32(kR32Tpo + ((nb + k) & (kTpRing - 1))) = det;
        }
      L64(kR64OutSq) = out_sq;
      L32(kR32OutPeak) = out_peak;
      L32(kR32NonFinite) = nonfinite;
      if (last_in_block && writer && row) {
        row->output_square_sum = out_sq;
        row->output_sample_peak = out_peak;
        row->non_finite_output = nonfinite != 0.0f ? 1u : 0u;
        row->tp_limiter_input_peak = tp_in_peak;
        row->tp_limiter_gr_db =
            (flags & kFlagLimiter) && tp_gmin < 1.0f ? -20.0f * log10f(fmaxf(tp_gmin, 1e-10f)) : 0.0f;
        row->tp_limited_events = tp_limited != 0.0f ? 1u : 0u;
      }
    }
    token_pass(turn, kTokTp, q);

    // ---- feed-forward, own sample: store, output-side 4x true peak (the detector of block_processor.rs:159)
    if (valid && own) {
      const float o = pick(x, kq);
      if (a.layout == 0) a.out[(int64_t)s * a.stream_stride + t0 + kq] = o;
      else a.out[(t0 + kq) * a.stream_stride + s] = o;
    }
    float otp;
    {
      float o4[kChunk];
      gather(own ? tp_observe_ring(&l32[kR32Tpo * kStreams], nb + kq, sl) : 0.0f, o4);
      otp = fmaxf(fmaxf(o4[0], o4[1]), fmaxf(o4[2], o4[3]));
    }
    // ---- token: fold the chunk's output true peak into the block maximum
    token_wait(turn, kTokFin, q);
    {
      const float m = fmaxf(first_in_block ? 0.0f : L32(kR32OutTp), otp);
      L32(kR32OutTp) = m;
      if (last_in_block && writer && row) row->output_true_peak = m;
    }
    token_pass(turn, kTokFin, q);
  }
  __syncthreads();
  if (tid == 0 && turn[kAbortSlot] != 0 && a.status) atomicExch(a.status, 1);

  // ---------------- write the state back (one lane per quad)
  if (writer) {
    struct Map { int row, field; };
    // the front-end rows (DC block, 80 Hz high-pass) are written back only by the launch that runs the front end:
    // with the suppressor on they belong to supp_prefilter_kernel, which may already be working on the next window
    const Map m64[] = {{kR64ScPrevIn, kCompScPrevIn},
                       {kR64ScPrevOut, kCompScPrevOut}, {kR64LowEnv, kCompLowEnv}, {kR64VoicedEnv, kCompVoicedEnv},
                       {kR64PresenceEnv, kCompPresenceEnv}, {kR64Plosive, kCompPlosive}, {kR64PeakEnvDb, kCompPeakEnvDb},
                       {kR64RmsEnvSq, kCompRmsEnvSq}, {kR64Gr, kCompGr}, {kR64FastEnv, kCompFastEnv},
                       {kR64SlowEnv, kCompSlowEnv}, {kR64CurReleaseMs, kCompCurReleaseMs},
                       {kR64TargetReleaseMs, kCompTargetReleaseMs}, {kR64SmoothedMakeup, kCompSmoothedMakeup},
                       {kR64LimGain, kLimGain}};
    const int n_m64 = (int)(sizeof(m64) / sizeof(m64[0]));
    for (int k = wave; k < n_m64; k += kRingWaves) a.st64[(int64_t)m64[k].field * NS + s] = L64(m64[k].row);
    for (int k = wave; k < (any_xf ? 4 : 2) * nsec; k += kRingWaves) {
      const int sec = k >> (any_xf ? 2 : 1), part = k & (any_xf ? 3 : 1);
      const int row = part < 2 ? kR64Eq + 2 * sec + part : pz_base + 2 * sec + (part - 2);
      a.st64[(int64_t)(kEqBase + 4 * sec + part) * NS + s] = L64(row);
    }
    if (wave == 0) {
      const double tau = fmax(L64(kR64CurReleaseMs), 0.001) / 1000.0;  // compressor.rs:760-761
      a.st64[(int64_t)kCompReleaseCoeff * NS + s] =
          P.comp.adaptive_release ? exp(-1.0 / (tau * P.comp.sample_rate)) : L64(kR64ReleaseCoeff);
      if (!(flags & kFlagCompressor)) a.st64[(int64_t)kCompGr * NS + s] = 0.0;
      if (flags & kFlagDcBlock) {
        a.st32[(int64_t)kDcX1 * NS + s] = L32(kR32DcX1);
        a.st32[(int64_t)kDcY1 * NS + s] = L32(kR32DcY1);
        a.st64[(int64_t)kPreZ1 * NS + s] = L64(kR64PreZ1);
        a.st64[(int64_t)kPreZ2 * NS + s] = L64(kR64PreZ2);
      }
      a.st32[(int64_t)kTpGain * NS + s] = L32(kR32TpGain);
      a.st32[(int64_t)kLimPrefix * NS + s] = L32(kR32LimPrefix);
    }
    const int64_t n_end = n0 + a.n_samples;
    for (int r = wave; r < kTpTaps; r += kRingWaves) {
      const int rowi = (int)((n_end - kTpTaps + r) & (kTpRing - 1));
      a.st32[(int64_t)(kTpInHist + r) * NS + s] = L32(kR32Tpi + rowi);
      a.st32[(int64_t)(kTpOutHist + r) * NS + s] = L32(kR32Tpo + rowi);
    }
    if (flags & kFlagLimiter)
      for (int r = wave; r < 2 * W; r += kRingWaves) a.st32[(int64_t)(kLimRing + r) * NS + s] = L32(kR32LimRing + r);
  }
#undef L64
#undef L32
}

size_t quad_kernel_dynamic_lds(int n_sections, int lookahead_samples, bool crossfade) {
  return quad::lds_bytes(n_sections, lookahead_samples, crossfade);
}

template <int kWaves>
static hipError_t launch_quad_variant(const LaunchArgs &args, size_t dyn, hipStream_t stream) {
  const int groups = (args.n_streams + quad::kStreams - 1) / quad::kStreams;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(chain_quad_kernel<kWaves>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) return err;
    attr_set = true;
  }
  hipLaunchKernelGGL((chain_quad_kernel<kWaves>), dim3(groups), dim3(kWaves * kLanes), dyn, stream, args);
  return hipGetLastError();
}

// `waves` = wavefronts per workgroup (ring depth): 16 (default), 12 or 8
hipError_t launch_chain_quad(const LaunchArgs &args, int n_sections, int lookahead_samples, bool crossfade, int waves,
                             hipStream_t stream) {
  const size_t dyn = quad::lds_bytes(n_sections, lookahead_samples, crossfade);
  switch (waves) {
    case 8: return launch_quad_variant<8>(args, dyn, stream);
    case 12: return launch_quad_variant<12>(args, dyn, stream);
    default: return launch_quad_variant<16>(args, dyn, stream);
  }
}

}  // namespace af
