// af_ring_kernel.hip -- Kernel 2: chain_ring_kernel ("token ring")
//
// Problem with lane-per-stream: 4096 streams are only 64 wavefronts, so 3/4 of the CUs idle
// and each busy SIMD runs one latency-bound wave.  The chain cannot be split along time
// (every stage is a recurrence), but it CAN be split along the stage graph:
//
//   * A workgroup of 16 wavefronts owns 64 streams; in every wave lane i is stream i, so
//     all code is uniform and every lane is busy.
//   * Time is cut into chunks of 4 samples.  Wave w processes chunks w, w+16, w+32, ... and
//     runs the WHOLE chain for its chunk, keeping the chunk's intermediates in registers.
//   * Each recurrence (EQ section group, side-chain filters, peak envelope, gain-reduction
//     smoothing, limiter, true-peak gain, ...) is a "serial unit" whose per-stream state sits
//     in LDS and is guarded by a token: the wave holding chunk q may run unit u only after
//     chunk q-1 has run it (turn[u] == q), then passes the token (turn[u] = q+1).
//   * Everything that is feed-forward (log10/exp10/sqrt/div of the compressor, both 4x
//     true-peak FIRs, loads/stores) runs outside the tokens, concurrently on all 16 waves.
//
// Every stream still sees its samples strictly in order through every recurrence, with the
// reference's operation order and rounding points, so results are identical to kernel 1.
// The longest token (five biquad sections x 4 samples) is ~1/16 of a chunk's total work, so
// in steady state the waves stagger themselves and nobody waits.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <type_traits>

#include "af_dsp.h"

namespace af {

constexpr int kTpRing = 128;  // rows of the shared true-peak input/output rings (power of two)
constexpr int kEqGroup = 5;   // biquad sections per token
constexpr int kMaxEqGroups = kMaxEqSections / kEqGroup;

// tokens
enum : int { kTokIn = 0, kTokCompA, kTokCompC, kTokCompE, kTokMeter, kTokLim, kTokTp, kTokFin, kTokEq0, kNumTokens = kTokEq0 + kMaxEqGroups };

// LDS rows, f64 plane (each row = 64 doubles)
enum : int {
  kR64PreZ1 = 0, kR64PreZ2,
  kR64ScPrevIn, kR64ScPrevOut, kR64LowEnv, kR64VoicedEnv, kR64PresenceEnv, kR64Plosive,
  kR64PeakEnvDb, kR64RmsEnvSq, kR64Gr, kR64FastEnv, kR64SlowEnv, kR64CurReleaseMs, kR64TargetReleaseMs,
  kR64ReleaseCoeff, kR64SmoothedMakeup, kR64MakeupLin,
  kR64LimGain, kR64LimGmin, kR64InSq, kR64OutSq,
  // auto-makeup: per-block activity estimate (double-buffered by block parity), meter filter, controller state
  kR64Activity0, kR64Activity1, kR64Reliab0, kR64Reliab1, kR64MeterV1, kR64MeterV2, kR64MeterV3, kR64MeterV4,
  kR64MeterAcc, kR64ActScore, kR64ActReliab, kR64CurrentLufs,
  kR64Eq  // then 2 rows per section (z1 z2), followed by 2 more per section (pz1 pz2) while a crossfade is pending
};
// LDS rows, f32 plane (each row = 64 floats)
enum : int {
  kR32DcX1 = 0, kR32DcY1, kR32TpGain, kR32LimPrefix, kR32InPeak, kR32OutPeak, kR32TpInPeak, kR32OutTp,
  kR32TpGmin, kR32TpLimited, kR32NonFinite, kR32Pad,
  kR32Tpi,                     // kTpRing rows
  kR32Tpo = kR32Tpi + kTpRing, // kTpRing rows
  kR32LimRing = kR32Tpo + kTpRing  // 2*W rows
};

__host__ __device__ inline size_t ring_lds_bytes(int n_sections, int lookahead, bool crossfade) {
  const size_t rows64 = kR64Eq + (crossfade ? 4 : 2) * (size_t)n_sections;
  const size_t rows32 = kR32LimRing + 2 * ((size_t)lookahead + 1);
  return 256 + rows64 * kLanes * sizeof(double) + rows32 * kLanes * sizeof(float);
}

// Every wave reaches the end of the kernel even if a token never arrives: after ~2^25 polls
// (seconds) the workgroup-wide abort word is raised, every later wait falls through, and the
// host reports the launch as failed instead of the GPU hanging.
// A feed-forward result pinned where it is written: LLVM may sink the arithmetic that produces a value used only inside a serial
// unit past the unit's token wait (earlier loads and pure arithmetic may legally move below an acquire), where it lengthens the
// time the token is held -- the kernel's period is its longest unit.  -DAF_NO_FF_PINS builds without (A/B).
// AF_PIN_MASK (A/B builds): 1 = in front of the peak-envelope unit, 2 = gain-reduction smoothing, 4 = limiter, 8 = true-peak
// limiter, 16 = final fold, 32 = the linear gains in front of the auto-makeup build's meter unit.
#ifndef AF_PIN_MASK
#define AF_PIN_MASK (2 | 16 | 32)
#endif
#define AF_PIN(bit, v) do { if constexpr (((AF_PIN_MASK) & (bit)) != 0) asm volatile("" : "+v"(v)); } while (0)
constexpr int kAbortSlot = 32;
constexpr int kPatienceSlot = 33;  // != 0: the launch follows a ready counter; a token may then be away for as long as that wait is allowed to last
#ifdef AF_TOKEN_PROFILE  // development aid (make EXTRA=-DAF_TOKEN_PROFILE): cycles every serial unit is waited for / held,
                         // printed by workgroup 0 at the end of each launch
__shared__ unsigned g_prof[2][16][12];
__shared__ long long g_prof_acq[16];
__shared__ long long g_prof_ready[16];
#endif
__device__ __forceinline__ void token_wait(int *turn_base, int tok, int q) {
  __builtin_amdgcn_sched_barrier(0);
#ifdef AF_TOKEN_PROFILE
  const long long prof_t0 = clock64();
#endif
  int spins = 0;
  while (__hip_atomic_load(&turn_base[tok], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != q) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 0xfff) == 0) {
      if (__hip_atomic_load(&turn_base[kAbortSlot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return;
      if (spins > (turn_base[kPatienceSlot] ? (1 << 28) : (1 << 25))) {
        __hip_atomic_store(&turn_base[kAbortSlot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
      }
    }
  }
  // a wave inside a serial unit is on the workgroup's critical path: it issues ahead of the three feed-forward waves
  // that share its SIMD (measured: 257 -> 240 ms of chain time per bench step)
  __builtin_amdgcn_s_setprio(3);
#ifdef AF_TOKEN_PROFILE
  if ((threadIdx.x & 63) == 0) {
    const long long t1 = clock64();
    g_prof[0][threadIdx.x >> 6][tok] += (unsigned)(t1 - prof_t0);
    g_prof_acq[threadIdx.x >> 6] = t1;
  }
#endif
  __builtin_amdgcn_sched_barrier(0);  // nothing that could have run before the wait is scheduled into the serial unit
}
__device__ __forceinline__ void token_pass(int *turn_base, int tok, int q) {
  __builtin_amdgcn_sched_barrier(0);  // ... and nothing that can wait until after the hand-over delays it
#ifdef AF_TOKEN_PROFILE
  if ((threadIdx.x & 63) == 0) g_prof[1][threadIdx.x >> 6][tok] += (unsigned)(clock64() - g_prof_acq[threadIdx.x >> 6]);
#endif
  __hip_atomic_store(&turn_base[tok], q + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  // Between serial units a wave keeps a priority that grows with the age of its chunk (a later stage means an older
  // chunk, and every unit is first-come-first-served, so the oldest chunk gates all others): chain time 237 -> 216 ms
  // per bench step on top of the in-unit priority (four levels; with three 219 ms; making the in-unit priority follow the
  // stage as well: 245 ms).
#ifndef AF_PRIO_SCHEME
#define AF_PRIO_SCHEME 0
#endif
#if AF_PRIO_SCHEME == 0
  if (tok == kTokEq0 + 1 || tok == kTokCompA) __builtin_amdgcn_s_setprio(1);
  else if (tok == kTokCompC || tok == kTokCompE) __builtin_amdgcn_s_setprio(2);
  else if (tok == kTokLim || tok == kTokTp) __builtin_amdgcn_s_setprio(3);
  else __builtin_amdgcn_s_setprio(0);
#elif AF_PRIO_SCHEME == 1  // (A/B builds) one level up from the compressor's first unit on
  if (tok == kTokEq0 + 1 || tok == kTokCompA) __builtin_amdgcn_s_setprio(2);
  else if (tok == kTokCompC || tok == kTokCompE) __builtin_amdgcn_s_setprio(3);
  else if (tok == kTokLim || tok == kTokTp) __builtin_amdgcn_s_setprio(3);
  else __builtin_amdgcn_s_setprio(0);
#elif AF_PRIO_SCHEME == 2  // flat: 1 everywhere between units
  if (tok == kTokFin) __builtin_amdgcn_s_setprio(0);
  else __builtin_amdgcn_s_setprio(1);
#elif AF_PRIO_SCHEME == 3  // steeper at the front: the chunk's first feed-forward zone (loads) at 1 already
  if (tok == kTokEq0 + 1 || tok == kTokCompA) __builtin_amdgcn_s_setprio(2);
  else if (tok == kTokCompC) __builtin_amdgcn_s_setprio(2);
  else if (tok == kTokCompE || tok == kTokLim || tok == kTokTp) __builtin_amdgcn_s_setprio(3);
  else __builtin_amdgcn_s_setprio(1);
#endif
  __builtin_amdgcn_sched_barrier(0);
}

// Bandlimited4xPeak::observe (true_peak.rs:173-186) over a shared ring: sample n sits in row n & 127
__device__ __forceinline__ float tp_observe_ring(const float *ring, int n, int lane) {
  float h[kTpTaps];
#pragma unroll
  for (int k = 0; k < kTpTaps; ++k) h[k] = ring[((n - k) & (kTpRing - 1)) * kLanes + lane];
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

// kAuto: the compressor's auto-makeup controller and its loudness meter are compiled in.
// (Round 1 also built this kernel cut in two launches at the compressor's static gain-reduction target, head and tail on
// different CUs a window apart: bit-identical, but each half stayed as long as the whole -- DESIGN.md 4.2c; removed.)
template <int kRingWaves, int kChunk, bool kAuto>
__global__ __launch_bounds__(kRingWaves *kLanes) void chain_ring_kernel(LaunchArgs a, const ChainParams *__restrict__ params) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // The parameter block is a kernel argument of its own, `const __restrict__`: nothing this kernel stores can alias it, so
  // its fields are uniform scalar loads (through the scalar cache).  As a pointer inside LaunchArgs every field read was a
  // global load behind an s_waitcnt vmcnt(0), most of them inside serial units (204 -> 37 global loads in the ISA).
  const ChainParams &P = params[a.group_preset ? a.group_preset[blockIdx.x] : 0];  // the preset of this 64-stream group
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
  float *l32 = reinterpret_cast<float *>(lds_raw + 256 + (size_t)rows64 * kLanes * sizeof(double));
#define L64(row) l64[(row)*kLanes + lane]
#define L32(row) l32[(row)*kLanes + lane]

  const bool comp_only = (flags & kFlagCompOnly) != 0;  // the launch ends at the compressor's output (the host strips kFlagLimiter too)
  const bool out_detector = !(flags & kFlagPrePass) && !comp_only;  // this launch runs the output-side detector
  const int tid = threadIdx.x;
  const int lane = tid & (kLanes - 1);
  const int wave = tid / kLanes;
  const int s0 = blockIdx.x * kLanes;
  const int s = s0 + lane;
  const bool valid = s < a.n_streams;
  const int sc = valid ? s : a.n_streams - 1;
  const int64_t NS = a.n_streams;
  const int64_t n0 = a.samples_before;  // absolute index of this launch's first sample

  // ---------------- stage the per-stream state into LDS (wave w takes rows w, w+16, ...)
  if (tid < 64) turn[tid] = (tid == kPatienceSlot && a.ready) ? 1 : 0;
#ifdef AF_TOKEN_PROFILE
  for (int i = tid; i < 2 * 16 * 12; i += kRingWaves * kLanes) (&g_prof[0][0][0])[i] = 0;
  if (tid < 16) g_prof_ready[tid] = 0;
  const long long prof_k0 = clock64(), prof_w0 = wall_clock64();
#endif
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
    if (wave == 1 % kRingWaves) {
      const int mbase = kF64Fixed + 4 * P.n_eq_sections;
      const bool meter = kAuto && P.comp.meter_slots > 0;
      L64(kR64MeterV1) = meter ? a.st64[(int64_t)(mbase + kMeterV1) * NS + sc] : 0.0;
      L64(kR64MeterV2) = meter ? a.st64[(int64_t)(mbase + kMeterV2) * NS + sc] : 0.0;
      L64(kR64MeterV3) = meter ? a.st64[(int64_t)(mbase + kMeterV3) * NS + sc] : 0.0;
      L64(kR64MeterV4) = meter ? a.st64[(int64_t)(mbase + kMeterV4) * NS + sc] : 0.0;
      L64(kR64MeterAcc) = 0.0;
      L64(kR64ActScore) = a.st64[(int64_t)kCompActivityScore * NS + sc];
      L64(kR64ActReliab) = a.st64[(int64_t)kCompActivityReliability * NS + sc];
      L64(kR64CurrentLufs) = a.st64[(int64_t)kCompCurrentLufs * NS + sc];
      L64(kR64Activity0) = 0.0;
      L64(kR64Activity1) = 0.0;
      L64(kR64Reliab0) = 0.0;
      L64(kR64Reliab1) = 0.0;
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
      L32(kR32Tpo + row) = out_detector ? a.st32[(int64_t)(kTpOutHist + r) * NS + sc] : 0.0f;
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
  const bool vec_ok = a.layout == 0 && (cb % kChunk) == 0 && (a.stream_stride % 4) == 0 && (kChunk % 2) == 0 &&
                      ((reinterpret_cast<uintptr_t>(a.in) | reinterpret_cast<uintptr_t>(a.out)) & 15) == 0;

  constexpr int64_t kAllReady = 0x7fffffffffffffffLL;
  int64_t ready_seen = a.ready ? 0 : kAllReady;  // (wave-uniform)
  for (int64_t q64 = wave; q64 < Q; q64 += kRingWaves) {
    const int q = (int)q64;
    const int64_t b = q64 / cpb;
    const int i = (int)(q64 - b * cpb);
    const int blk_len = (int)((a.n_samples - b * cb) < cb ? (a.n_samples - b * cb) : cb);
    const int64_t t0 = b * cb + (int64_t)i * kChunk;  // launch-relative index of the chunk's first sample
    const int len = (blk_len - i * kChunk) < kChunk ? (blk_len - i * kChunk) : kChunk;
    if (t0 + len > ready_seen) {
      // The launch follows its producers (a.ready): the chunk's samples -- and the block's pre-pass power -- are there once the
      // counter covers them.  The wave holds no token here, so the waves ahead of it drain on.  Polls are relaxed agent-scope
      // loads; ONE acquire fence after the last one makes the producers' writes (other XCDs: written back at their kernels' ends,
      // the counter published after that) visible to this wave's loads.  Bounded: a counter that never arrives ends the wait after
      // ~4 s, reports through the status word and lets every wave run to the end of the launch.
      int spins = 0;
#ifdef AF_TOKEN_PROFILE
      const long long prof_r0 = clock64();
#endif
      for (;;) {
        ready_seen = __hip_atomic_load(a.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ready_seen >= t0 + len) break;
        __builtin_amdgcn_s_sleep(32);
        if ((++spins & 0xff) == 0 &&
            (spins > (1 << 22) || __hip_atomic_load(&turn[kAbortSlot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0)) {
          __hip_atomic_store(&turn[kAbortSlot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          ready_seen = kAllReady;
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#ifdef AF_TOKEN_PROFILE
      if ((threadIdx.x & 63) == 0) g_prof_ready[threadIdx.x >> 6] += clock64() - prof_r0;
#endif
    }
    const bool first_in_block = i == 0;
    const bool last_in_block = i * kChunk + len == blk_len;
    BlockStats *row = a.stats ? &a.stats[b * NS + sc] : nullptr;

    const int nb = (int)((n0 + t0) & (kTpRing - 1));  // ring row of the chunk's first sample
    auto chunk_body = [&](auto full_tag) {
      constexpr bool kFull = decltype(full_tag)::value;  // a full chunk needs no per-sample guards
      // ---- load the chunk (issued before the first token wait)
      float x[kChunk];
      if (vec_ok && kFull) {
        const float *src = &a.in[(int64_t)s * a.stream_stride + t0];
        if constexpr (kChunk == 2) {
          const float2 v = valid ? *reinterpret_cast<const float2 *>(src) : make_float2(0, 0);
          x[0] = v.x; x[1] = v.y;
        } else {
  #pragma unroll
          for (int k4 = 0; k4 < kChunk; k4 += 4) {
            const float4 v = valid ? *reinterpret_cast<const float4 *>(src + k4) : make_float4(0, 0, 0, 0);
            x[k4] = v.x; x[k4 + 1] = v.y; x[k4 + 2] = v.z; x[k4 + 3] = v.w;
          }
        }
      } else {
  #pragma unroll
        for (int k = 0; k < kChunk; ++k) {
          x[k] = 0.0f;
          if (valid && (kFull || k < len))
            x[k] = a.layout == 0 ? a.in[(int64_t)s * a.stream_stride + t0 + k] : a.in[(t0 + k) * a.stream_stride + s];
        }
      }

      double target[kChunk];

      // =========================== token: input scrub, block input stats, DC block + fixed HP
      if (!(flags & kFlagInputDone)) {
      token_wait(turn, kTokIn, q);
      {
        double in_sq = first_in_block ? 0.0 : L64(kR64InSq);
        float in_peak = first_in_block ? 0.0f : L32(kR32InPeak);
  #pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (kFull || k < len) {
            float v = x[k];
            if ((flags & (kFlagInputScrub | kFlagInputClamp)) && !finite_f32(v)) v = 0.0f;
            if (flags & kFlagInputClamp) v = fclamp(v, -1.0f, 1.0f);
            x[k] = v;
            in_sq += (double)v * (double)v;
            in_peak = fmaxf(in_peak, fabsf(v));
          }
        L64(kR64InSq) = in_sq;
        L32(kR32InPeak) = in_peak;
        if (last_in_block && valid && row) {
          row->input_square_sum = in_sq;
          row->input_sample_peak = in_peak;
        }
        if (flags & kFlagDcBlock) {  // routing.rs:826-843
          float dc_x1 = L32(kR32DcX1), dc_y1 = L32(kR32DcY1);
          double z1 = L64(kR64PreZ1), z2 = L64(kR64PreZ2);
          const BiquadCoef c = P.pre_hp;
  #pragma unroll
          for (int k = 0; k < kChunk; ++k)
            if (kFull || k < len) {
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
      }

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
              if (kFull || k < len) {
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
              if (kFull || k < len) {
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
        double d[kChunk], inst_peak_db[kChunk], rms_db[kChunk], weight_db[kChunk];
        double low_e[kChunk], voiced_e[kChunk], presence_e[kChunk], rms_e[kChunk];
        // ---- token A: side-chain high-pass + band / rms envelopes (linear recurrences)
        token_wait(turn, kTokCompA, q);
        {
          if (kAuto && first_in_block) {
            // estimate_auto_makeup_activity(block_rms_db(buffer), evidence), compressor.rs:528-596,710
            const double power = a.pre_power   ? a.pre_power[b * NS + sc] / (double)blk_len
                                 : a.pre_stats ? a.pre_stats[b * NS + sc].output_square_sum / (double)blk_len
                                               : 0.0;
            const double brms_db = lin2db(sqrt(power), 1e-10);
            double absolute = 0.0;
            if (brms_db >= -55.0 && brms_db <= -6.0)
              absolute = fmin(dclamp(div_known(brms_db + 55.0, 12.0, 1.0 / 12.0), 0.0, 1.0),
                              dclamp(div_known(-6.0 - brms_db, 6.0, 1.0 / 6.0), 0.0, 1.0));
            double act = absolute, rel = 1.0;
            if (cp.has_evidence) {
              double vad_rel = cp.vad_reliability;
              double vad_p = a.vad_prob ? a.vad_prob[b * NS + sc] : 0.0;
              if (!(fabs(vad_p) < HUGE_VAL) || vad_p != vad_p) {
                vad_rel = 0.0;
                vad_p = 0.0;
              }
              vad_p = dclamp(vad_p, 0.0, 1.0);
              const double configured = cp.noise_reference_reliability;
              const double live = cp.live_noise_reliability;
              double noise_rel = configured > 0.0 ? fmin(live, configured) : live;
              double relative = 0.0;
              const double nf = cp.noise_floor_db;
              if (nf >= -120.0 && nf <= 0.0) {
                const double e0 = nf + 3.0, e1 = nf + 15.0;
                const double t = dclamp((brms_db - e0) / (e1 - e0), 0.0, 1.0);
                relative = t * t * (3.0 - 2.0 * t);
              } else {
                noise_rel = 0.0;
              }
              const double fallback = noise_rel * relative + (1.0 - noise_rel) * absolute;
              act = dclamp(vad_rel * vad_p + (1.0 - vad_rel) * fallback, 0.0, 1.0);
              rel = dclamp(fmax(vad_rel, 0.75 * noise_rel), 0.0, 1.0);
            }
            L64((b & 1) ? kR64Activity1 : kR64Activity0) = act;
            L64((b & 1) ? kR64Reliab1 : kR64Reliab0) = rel;
          }
          double rms_env = L64(kR64RmsEnvSq);
          if (cp.sidechain_highpass_enabled) {
            double prev_in = L64(kR64ScPrevIn), prev_out = L64(kR64ScPrevOut);
            double low_env = L64(kR64LowEnv), voiced_env = L64(kR64VoicedEnv), presence_env = L64(kR64PresenceEnv);
            const double kk = cp.band_env_coeff;
  #pragma unroll
            for (int k = 0; k < kChunk; ++k)
              if (kFull || k < len) {
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
              if (kFull || k < len) {
                const double dd = (double)x[k];
                rms_env = cp.rms_coeff * rms_env + (1.0 - cp.rms_coeff) * (dd * dd);
                d[k] = dd;
                rms_e[k] = rms_env;
              }
          }
          L64(kR64RmsEnvSq) = rms_env;
        }
        token_pass(turn, kTokCompA, q);
        // ---- feed-forward: detector weight, instantaneous peak and RMS levels in dB
        double plosive_last = 0.0;
  #pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (kFull || k < len) {
            weight_db[k] = kDetectorUnitWeight;  // (weight_db / rms_db: dB in the literal build, linear otherwise -- af_dsp.h, detector_db)
            if (cp.sidechain_highpass_enabled) {  // update_sidechain_band_metrics, compressor.rs:438-449
              const double low_rms = sqrt(low_e[k]);
              const double voiced_rms = fmax(sqrt(voiced_e[k]), 1e-8);
              const double presence_rms = sqrt(presence_e[k]);
              const double plosive = dclamp(low_rms / voiced_rms, 0.0, 32.0);
              plosive_last = plosive;
              const double plosive_amount = dclamp(div_known(plosive - 1.25, 3.75, 1.0 / 3.75), 0.0, 1.0);
              const double plosive_penalty = 1.0 - plosive_amount * (1.0 - 0.35);
              const double presence_ratio = dclamp(presence_rms / voiced_rms, 0.0, 4.0);
              const double presence_weight = 1.0 + 0.18 * dclamp(presence_ratio - 0.75, 0.0, 1.0);
              weight_db[k] = detector_weight(dclamp(plosive_penalty * presence_weight, 0.35, 1.15));
            }
            inst_peak_db[k] = lin2db(fabs(d[k]), 1e-10);
            rms_db[k] = detector_rms_level(rms_e[k]);
          }
        // ---- token C: log-domain peak envelope (compressor.rs:735-742)
        double peak_db[kChunk];
  #pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (kFull || k < len) AF_PIN(1, inst_peak_db[k]);
        token_wait(turn, kTokCompC, q);
        {
          double pe = L64(kR64PeakEnvDb);
  #pragma unroll
          for (int k = 0; k < kChunk; ++k)
            if (kFull || k < len) {
              const double pk = inst_peak_db[k] > pe ? cp.attack_coeff : cp.detector_release_coeff;
              pe = pk * pe + (1.0 - pk) * inst_peak_db[k];
              peak_db[k] = pe;
            }
          L64(kR64PeakEnvDb) = pe;
          if (len > 0) L64(kR64Plosive) = plosive_last;  // diagnostic state only (compressor.rs:441)
        }
        token_pass(turn, kTokCompC, q);
        // ---- feed-forward: blended detector level -> static gain-reduction target
  #pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (kFull || k < len) {
            target[k] = comp_gain_reduction(cp, detector_db(peak_db[k], rms_db[k], weight_db[k]));
          }
        // ---- token E: release-time meter + gain-reduction smoothing (compressor.rs:452-505,752-764)
        double gr_k[kChunk];
        double makeup_lin;
  #pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (kFull || k < len) AF_PIN(2, target[k]);
        token_wait(turn, kTokCompE, q);
        {
          double gr = L64(kR64Gr), fast = L64(kR64FastEnv), slow = L64(kR64SlowEnv);
          double cur_ms = L64(kR64CurReleaseMs), tgt_ms = L64(kR64TargetReleaseMs);
          const double rel_coeff = L64(kR64ReleaseCoeff);
          makeup_lin = L64(kR64MakeupLin);
  #pragma unroll
          for (int k = 0; k < kChunk; ++k)
            if (kFull || k < len) {
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
          if (last_in_block && valid && row) row->compressor_gr_db = (float)gr;
          if (last_in_block && !kAuto) {  // update_auto_makeup_gain, auto-makeup off (compressor.rs:604-617)
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
            if (valid && row) row->makeup_gain_db = (float)sm;
          }
        }
        token_pass(turn, kTokCompE, q);
        if constexpr (!kAuto) {
          // ---- feed-forward: apply gain (compressor.rs:771-773)
  #pragma unroll
          for (int k = 0; k < kChunk; ++k)
            if (kFull || k < len) x[k] = (float)((double)x[k] * (db2lin(-gr_k[k]) * makeup_lin));
        } else {
          double glin[kChunk];
  #pragma unroll
          for (int k = 0; k < kChunk; ++k)
            if (kFull || k < len) {
              glin[k] = db2lin(-gr_k[k]);
              AF_PIN(32, glin[k]);
            }
          // ---- token: makeup gain of THIS block, loudness meter, auto-makeup controller at block end
          token_wait(turn, kTokMeter, q);
          {
            const double mk = L64(kR64MakeupLin);
            const double act = L64((b & 1) ? kR64Activity1 : kR64Activity0);
            const double rel = L64((b & 1) ? kR64Reliab1 : kR64Reliab0);
            const bool fed = act > 0.20 && rel >= 0.35 && cp.meter_slots > 0;  // compressor.rs:714-720
            double v1 = L64(kR64MeterV1), v2 = L64(kR64MeterV2), v3 = L64(kR64MeterV3), v4 = L64(kR64MeterV4);
            double acc = first_in_block ? 0.0 : L64(kR64MeterAcc);
  #pragma unroll
            for (int k = 0; k < kChunk; ++k)
              if (kFull || k < len) {
                x[k] = (float)((double)x[k] * (glin[k] * mk));
                if (fed) {  // K-weighting, one 4th-order direct-form section (loudness.rs:119-127 over ebur128)
                  const double v0 = (double)x[k] - cp.kw_a[1] * v1 - cp.kw_a[2] * v2 - cp.kw_a[3] * v3 - cp.kw_a[4] * v4;
                  const double y = cp.kw_b[0] * v0 + cp.kw_b[1] * v1 + cp.kw_b[2] * v2 + cp.kw_b[3] * v3 + cp.kw_b[4] * v4;
                  v4 = v3; v3 = v2; v2 = v1; v1 = v0;
                  acc += y * y;
                }
              }
            if (last_in_block) {
              const int mbase = kF64Fixed + 4 * P.n_eq_sections;
              double lufs = L64(kR64CurrentLufs);
              if (fed) {
                const double tiny = 2.2250738585072014e-308;
                if (fabs(v1) < tiny) v1 = 0.0;
                if (fabs(v2) < tiny) v2 = 0.0;
                if (fabs(v3) < tiny) v3 = 0.0;
                if (fabs(v4) < tiny) v4 = 0.0;
                // 400 ms window = the last meter_slots fed blocks (block energies instead of 19 200 samples)
                // (the slots live in HBM: written by whichever wave ends a block, so they are read and
                // written with agent-scope accesses that bypass this CU's L1)
                const int written = (int)__hip_atomic_load(&a.st64[(int64_t)(mbase + kMeterPos) * NS + sc], __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
                const int pos = written + 1 == cp.meter_slots ? 0 : written + 1;
                if (valid) {
                  __hip_atomic_store(&a.st64[(int64_t)(mbase + kMeterRing + written) * NS + s], acc, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
                  __hip_atomic_store(&a.st64[(int64_t)(mbase + kMeterPos) * NS + s], (double)pos, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
                }
                double sum = 0.0;
                for (int m = 0; m < cp.meter_slots; ++m) {
                  const double e = m == written ? acc
                                                : __hip_atomic_load(&a.st64[(int64_t)(mbase + kMeterRing + m) * NS + sc],
                                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  sum += e;
                }
                const double energy = sum / cp.meter_frames;
                lufs = energy <= 0.0 ? -HUGE_VAL : (double)(float)(10.0 * (log(energy) / log(10.0)) - 0.691);
                L64(kR64CurrentLufs) = lufs;
              }
              // update_auto_makeup_gain, compressor.rs:598-653
              const bool whole = blk_len == P.control_block;
              const double elapsed = (double)(blk_len < 1 ? 1 : blk_len);
              const double mc = whole ? cp.makeup_pow_cb : pow(cp.makeup_smoothing_coeff, elapsed);
              const double rc2 = whole ? cp.relax_pow_cb : pow(cp.makeup_silence_relax_coeff, elapsed);
              const double ac = whole ? cp.activity_pow_cb : pow(cp.speech_activity_smoothing_coeff, elapsed);
              double sm = L64(kR64SmoothedMakeup);
              const double score = ac * L64(kR64ActScore) + (1.0 - ac) * dclamp(act, 0.0, 1.0);
              const double relst = dclamp(rel, 0.0, 1.0);
              L64(kR64ActScore) = score;
              L64(kR64ActReliab) = relst;
              if (score < 0.20) {
                sm = rc2 * sm + (1.0 - rc2) * cp.makeup_gain_db;
              } else if (relst < 0.35) {
                const double cap = cp.makeup_gain_db + 3.0 * (relst / 0.35);
                if (sm > cap) sm = mc * sm + (1.0 - mc) * cap;
              } else {
                const double required = cp.target_lufs - lufs;
                const double reliability_cap = dclamp(12.0 * relst, 3.0, 12.0);
                const double headroom_cap = dclamp(12.0, 0.0, reliability_cap);  // limiter feedback is 0 offline
                const double clamped = dclamp(required, 0.0, headroom_cap);
                if (fabs(clamped - sm) > 0.1) {
                  sm = mc * sm + (1.0 - mc) * clamped;
                } else {
                  sm = clamped;
                }
              }
              L64(kR64SmoothedMakeup) = sm;
              L64(kR64MakeupLin) = db2lin(sm);
              if (valid && row) {
                row->makeup_gain_db = (float)sm;
                row->makeup_activity = (float)score;
                row->makeup_reliability = (float)relst;
              }
            }
            L64(kR64MeterV1) = v1;
            L64(kR64MeterV2) = v2;
            L64(kR64MeterV3) = v3;
            L64(kR64MeterV4) = v4;
            L64(kR64MeterAcc) = acc;
          }
          token_pass(turn, kTokMeter, q);
        }
      }

      // =========================== limiter + true-peak limiter
      float itp[kChunk];
      if (flags & kFlagLimiter) {
        const double ceil_lin = P.lim.ceiling_linear;
        const double rc = P.lim.release_coeff;
        float *ring = &l32[kR32LimRing * kLanes];
        float *suf = &l32[(kR32LimRing + W) * kLanes];
        // ---- token: lookahead limiter (limiter.rs:246-284), sliding max by block prefix/suffix maxima
  #pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (kFull || k < len) AF_PIN(4, x[k]);  // (the compressor's gain applied: an exp10 per sample)
        token_wait(turn, kTokLim, q);
        {
          double g = L64(kR64LimGain);
          double gmin = first_in_block ? 1.0 : L64(kR64LimGmin);
          float prefix = L32(kR32LimPrefix);
          int j = (int)((n0 + t0) % W);
  #pragma unroll
          for (int k = 0; k < kChunk; ++k)
            if (kFull || k < len) {
              const float xin = x[k];
              const float ax = fabsf(xin);
              const int jn = (j + 1 == W) ? 0 : j + 1;
              const float delayed = ring[jn * kLanes + lane];
              const float sfx = (j + 1 < W) ? suf[(j + 1) * kLanes + lane] : 0.0f;
              prefix = (j == 0) ? ax : fmaxf(prefix, ax);
              const double peak = (double)fmaxf(sfx, prefix);
              ring[j * kLanes + lane] = xin;
              if (j + 1 == W) {
                // suffix maxima of the block just completed, eight at a time: the loads of a batch are issued together (as one
                // load -> max -> store per element the compiler must assume `suf` aliases `ring` and pays an LDS round trip per
                // element, W of them inside the serial unit every W samples)
                float m = 0.0f;
                int kk = W - 1;
                for (; kk >= 7; kk -= 8) {
                  float v[8];
  #pragma unroll
                  for (int u = 0; u < 8; ++u) v[u] = fabsf(ring[(kk - u) * kLanes + lane]);
  #pragma unroll
                  for (int u = 0; u < 8; ++u) {
                    m = fmaxf(m, v[u]);
                    suf[(kk - u) * kLanes + lane] = m;
                  }
                }
                for (; kk >= 0; --kk) {
                  m = fmaxf(m, fabsf(ring[kk * kLanes + lane]));
                  suf[kk * kLanes + lane] = m;
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
          if (last_in_block && valid && row)
            row->limiter_peak_gr_db = gmin < 1.0 ? (float)(-lin2db(gmin, 1e-10)) : 0.0f;
        }
        token_pass(turn, kTokLim, q);
        // ---- feed-forward: input-side 4x true peak
  #pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (kFull || k < len) {
            itp[k] = tp_observe_ring(&l32[kR32Tpi * kLanes], nb + k, lane);
            AF_PIN(8, itp[k]);
          }
      }

      // ---- token: true-peak gain (true_peak.rs:341-374), chain output, block output stats
      if (!comp_only) {
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
            if (kFull || k < len) {
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
          if (kFull || k < len) {
            const float o = x[k];
            float det = o;
            if (finite_f32(o)) {
              out_sq += (double)o * (double)o;
            } else {
              nonfinite = 1.0f;
              det = 0.0f;  // TruePeakDetector::process_block, true_peak.rs:212
            }
            out_peak = fmaxf(out_peak, fabsf(o));
            if (out_detector) L32(kR32Tpo + ((nb + k) & (kTpRing - 1))) = det;
          }
        L64(kR64OutSq) = out_sq;
        L32(kR32OutPeak) = out_peak;
        L32(kR32NonFinite) = nonfinite;
        if (last_in_block && valid && row) {
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
      }

      // ---- feed-forward: store the chunk, output-side 4x true peak (the detector of block_processor.rs:159)
      if (vec_ok && kFull) {
        if (valid) {
          float *dst = &a.out[(int64_t)s * a.stream_stride + t0];
          if constexpr (kChunk == 2) {
            *reinterpret_cast<float2 *>(dst) = make_float2(x[0], x[1]);
          } else {
  #pragma unroll
            for (int k4 = 0; k4 < kChunk; k4 += 4)
              *reinterpret_cast<float4 *>(dst + k4) = make_float4(x[k4], x[k4 + 1], x[k4 + 2], x[k4 + 3]);
          }
        }
      } else {
  #pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (valid && (kFull || k < len)) {
            if (a.layout == 0) a.out[(int64_t)s * a.stream_stride + t0 + k] = x[k];
            else a.out[(t0 + k) * a.stream_stride + s] = x[k];
          }
      }
      float otp = 0.0f;
      if (out_detector) {
  #pragma unroll
        for (int k = 0; k < kChunk; ++k)
          if (kFull || k < len) otp = fmaxf(otp, tp_observe_ring(&l32[kR32Tpo * kLanes], nb + k, lane));
        // The fold below needs one number.  Left alone, LLVM sinks the 512 multiply-adds that produce it past the token's acquire
        // (legal: earlier loads may move below an acquire) and the LAST serial unit holds its token for ~3 000 cycles per chunk --
        // the longest unit of the kernel once the EQ has left it, i.e. its period.  Pinned here, the unit is a max and a store.
        AF_PIN(16, otp);
      }
      // ---- token: fold the chunk's output true peak into the block maximum
      if (out_detector) {
        token_wait(turn, kTokFin, q);
        {
          const float m = fmaxf(first_in_block ? 0.0f : L32(kR32OutTp), otp);
          L32(kR32OutTp) = m;
          if (last_in_block && valid && row) row->output_true_peak = m;
        }
        token_pass(turn, kTokFin, q);
      }
    };
    if (len == kChunk) chunk_body(std::true_type{});
    else chunk_body(std::false_type{});
  }
  __syncthreads();
  if (tid == 0 && turn[kAbortSlot] != 0 && a.status) atomicExch(a.status, 1);
#ifdef AF_TOKEN_PROFILE
  if (tid == 0 && blockIdx.x == 0 && Q > 0) {
    const long long ticks = clock64() - prof_k0, wall = wall_clock64() - prof_w0;  // shader cycles; 100 MHz constant clock
    printf("ring profile: %lld chunks, kernel %lld ticks = %.1f per chunk, %.3f ms wall, clock %.3f GHz\n", (long long)Q, ticks,
           (double)ticks / (double)Q, (double)wall * 1e-5, (double)ticks / ((double)wall * 10.0));
    if (a.ready) {
      long long r = 0;
      for (int k = 0; k < kRingWaves; ++k) r += g_prof_ready[k];
      printf("  waiting for the ready counter: %.3f ms per wave (wave 0: %.3f ms)\n",
             (double)r / kRingWaves / ((double)ticks / ((double)wall * 1e-5)), (double)g_prof_ready[0] / ((double)ticks / ((double)wall * 1e-5)));
    }
    for (int t = 0; t < kTokEq0 + n_groups; ++t) {
      long long w = 0, h = 0;
      for (int k = 0; k < kRingWaves; ++k) { w += g_prof[0][k][t]; h += g_prof[1][k][t]; }
      printf("  token %2d: held %.1f ticks per chunk, waited for %.1f\n", t, (double)h / (double)Q, (double)w / (double)Q);
    }
  }
#endif

  // ---------------- write the state back
  if (valid && (flags & kFlagPrePass)) {
    // pre-pass launch: only the front end and the EQ advanced
    for (int k = wave; k < (any_xf ? 4 : 2) * nsec; k += kRingWaves) {
      const int sec = k >> (any_xf ? 2 : 1), part = k & (any_xf ? 3 : 1);
      const int row = part < 2 ? kR64Eq + 2 * sec + part : pz_base + 2 * sec + (part - 2);
      a.st64[(int64_t)(kEqBase + 4 * sec + part) * NS + s] = L64(row);
    }
    if (wave == 0 && (flags & kFlagDcBlock)) {
      a.st64[(int64_t)kPreZ1 * NS + s] = L64(kR64PreZ1);
      a.st64[(int64_t)kPreZ2 * NS + s] = L64(kR64PreZ2);
      a.st32[(int64_t)kDcX1 * NS + s] = L32(kR32DcX1);
      a.st32[(int64_t)kDcY1 * NS + s] = L32(kR32DcY1);
    }
  } else if (valid) {
    struct Map { int row, field; };
    // the front-end rows (DC block, 80 Hz high-pass) are written back only by the launch that runs the front end:
    // with the suppressor on they belong to supp_prefilter_kernel, which may already be working on the next window
    const Map head64[] = {{kR64ScPrevIn, kCompScPrevIn}, {kR64ScPrevOut, kCompScPrevOut}, {kR64LowEnv, kCompLowEnv},
                          {kR64VoicedEnv, kCompVoicedEnv}, {kR64PresenceEnv, kCompPresenceEnv}, {kR64Plosive, kCompPlosive},
                          {kR64PeakEnvDb, kCompPeakEnvDb}, {kR64RmsEnvSq, kCompRmsEnvSq}};
    const Map tail64[] = {{kR64LimGain, kLimGain},  // (first: skipped when another kernel owns the limiter)
                          {kR64Gr, kCompGr}, {kR64FastEnv, kCompFastEnv}, {kR64SlowEnv, kCompSlowEnv},
                          {kR64CurReleaseMs, kCompCurReleaseMs}, {kR64TargetReleaseMs, kCompTargetReleaseMs},
                          {kR64SmoothedMakeup, kCompSmoothedMakeup},
                          {kR64ActScore, kCompActivityScore}, {kR64ActReliab, kCompActivityReliability},
                          {kR64CurrentLufs, kCompCurrentLufs}};
    constexpr int n_head64 = (int)(sizeof(head64) / sizeof(head64[0])), n_tail64 = (int)(sizeof(tail64) / sizeof(tail64[0]));
    {  // rows of the tokens up to the gain-reduction target
      for (int k = wave; k < n_head64; k += kRingWaves) a.st64[(int64_t)head64[k].field * NS + s] = L64(head64[k].row);
      for (int k = wave; k < (any_xf ? 4 : 2) * nsec; k += kRingWaves) {
        const int sec = k >> (any_xf ? 2 : 1), part = k & (any_xf ? 3 : 1);
        const int row = part < 2 ? kR64Eq + 2 * sec + part : pz_base + 2 * sec + (part - 2);
        a.st64[(int64_t)(kEqBase + 4 * sec + part) * NS + s] = L64(row);
      }
      if (wave == 0 && (flags & kFlagDcBlock)) {
        a.st32[(int64_t)kDcX1 * NS + s] = L32(kR32DcX1);
        a.st32[(int64_t)kDcY1 * NS + s] = L32(kR32DcY1);
        a.st64[(int64_t)kPreZ1 * NS + s] = L64(kR64PreZ1);
        a.st64[(int64_t)kPreZ2 * NS + s] = L64(kR64PreZ2);
      }
    }
    {  // rows of the tokens from the gain-reduction smoothing on
      for (int k = wave; k < n_tail64; k += kRingWaves)
        if (k > 0 || !comp_only) a.st64[(int64_t)tail64[k].field * NS + s] = L64(tail64[k].row);
      if (wave == 1 % kRingWaves && kAuto && P.comp.meter_slots > 0) {
        const int mbase = kF64Fixed + 4 * P.n_eq_sections;
        a.st64[(int64_t)(mbase + kMeterV1) * NS + s] = L64(kR64MeterV1);
        a.st64[(int64_t)(mbase + kMeterV2) * NS + s] = L64(kR64MeterV2);
        a.st64[(int64_t)(mbase + kMeterV3) * NS + s] = L64(kR64MeterV3);
        a.st64[(int64_t)(mbase + kMeterV4) * NS + s] = L64(kR64MeterV4);
      }
      if (wave == 0) {
        const double tau = fmax(L64(kR64CurReleaseMs), 0.001) / 1000.0;  // compressor.rs:760-761
        a.st64[(int64_t)kCompReleaseCoeff * NS + s] =
            P.comp.adaptive_release ? exp(-1.0 / (tau * P.comp.sample_rate)) : L64(kR64ReleaseCoeff);
        if (!(flags & kFlagCompressor)) a.st64[(int64_t)kCompGr * NS + s] = 0.0;
        if (!comp_only) {
          a.st32[(int64_t)kTpGain * NS + s] = L32(kR32TpGain);
          a.st32[(int64_t)kLimPrefix * NS + s] = L32(kR32LimPrefix);
        }
      }
      const int64_t n_end = n0 + a.n_samples;
      for (int r = wave; r < (comp_only ? 0 : kTpTaps); r += kRingWaves) {
        const int rowi = (int)((n_end - kTpTaps + r) & (kTpRing - 1));
        a.st32[(int64_t)(kTpInHist + r) * NS + s] = L32(kR32Tpi + rowi);
        if (out_detector) a.st32[(int64_t)(kTpOutHist + r) * NS + s] = L32(kR32Tpo + rowi);
      }
      if (flags & kFlagLimiter)
        for (int r = wave; r < 2 * W; r += kRingWaves) a.st32[(int64_t)(kLimRing + r) * NS + s] = L32(kR32LimRing + r);
    }
  }
#undef L64
#undef L32
}

size_t ring_kernel_dynamic_lds(int n_sections, int lookahead_samples, bool crossfade) {
  return ring_lds_bytes(n_sections, lookahead_samples, crossfade);
}

#ifdef AF_TOKEN_PROFILE
constexpr int kRingMaxDynamicLds = 160 * 1024 - 4096;  // the profile's static arrays come out of the same 160 KB
#else
constexpr int kRingMaxDynamicLds = 160 * 1024;
#endif
template <int kRingWaves, int kChunk, bool kAuto = false>
static hipError_t launch_variant(const LaunchArgs &args, size_t dyn, hipStream_t stream) {
  const int groups = (args.n_streams + kLanes - 1) / kLanes;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(chain_ring_kernel<kRingWaves, kChunk, kAuto>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kRingMaxDynamicLds);
    if (err != hipSuccess) return err;
    attr_set = true;
  }
  hipLaunchKernelGGL((chain_ring_kernel<kRingWaves, kChunk, kAuto>), dim3(groups), dim3(kRingWaves * kLanes), dyn, stream,
                     args, args.params);
  return hipGetLastError();
}

// `variant` = waves * 100 + chunk; 0 picks the default
hipError_t launch_chain_ring_lds(const LaunchArgs &args, size_t dyn, int variant, bool auto_makeup, hipStream_t stream);
hipError_t launch_chain_ring(const LaunchArgs &args, int n_sections, int lookahead_samples, bool crossfade, int variant,
                             bool auto_makeup, hipStream_t stream) {
  return launch_chain_ring_lds(args, ring_lds_bytes(n_sections, lookahead_samples, crossfade), variant, auto_makeup, stream);
}
// The producers' side of LaunchArgs::ready: enqueued on the producers' stream behind the kernel that completed the samples (its
// writes are written back when it ends), one thread publishes the new count with an agent-scope release store.
__global__ void chain_publish_ready_kernel(int64_t *ready, int64_t samples) {
  __hip_atomic_store(ready, samples, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
hipError_t launch_chain_publish_ready(int64_t *ready, int64_t samples, hipStream_t stream) {
  hipLaunchKernelGGL(chain_publish_ready_kernel, dim3(1), dim3(1), 0, stream, ready, samples);
  return hipGetLastError();
}

// `dyn`: dynamic LDS of the launch (several presets: the largest of their layouts, every workgroup lays out its own)
hipError_t launch_chain_ring_lds(const LaunchArgs &args, size_t dyn, int variant, bool auto_makeup, hipStream_t stream) {
  if (auto_makeup) {
    // AF_AUTO_WAVES=8|12|16 (same-box A/B): 8 waves hold everything in 235 VGPRs; 12 and 16 spill a little and keep more chunks
    // in flight (the kernel is a closed queueing network of its waves)
    static const int waves = [] {
      const char *env = std::getenv("AF_AUTO_WAVES");
      // full bench step with auto-makeup, same box: round 3's first build 8 -> 276 ms, 12 -> 228, 16 -> 232 (plain chain: 203); final
      // build (EQ off this kernel, one launch per call, the last unit freed of the detector's FIR): 8 -> 222.7, 12 -> 184.4, 16 -> 179.9
      return env ? std::atoi(env) : 16;
    }();
    if (waves == 8) return launch_variant<8, 4, true>(args, dyn, stream);
    if (waves == 12) return launch_variant<12, 4, true>(args, dyn, stream);
    return launch_variant<16, 4, true>(args, dyn, stream);
  }
  switch (variant) {
    case 1604: return launch_variant<16, 4>(args, dyn, stream);
    case 1602: return launch_variant<16, 2>(args, dyn, stream);
    case 804: return launch_variant<8, 4>(args, dyn, stream);
    case 802: return launch_variant<8, 2>(args, dyn, stream);
    case 1204: return launch_variant<12, 4>(args, dyn, stream);
    case 1202: return launch_variant<12, 2>(args, dyn, stream);
    case 0:
    default: return launch_variant<16, 4>(args, dyn, stream);
  }
}

}  // namespace af
