// af_eq_systolic.hip -- the 10-band EQ (ParametricEQ::process_block_inplace, eq.rs:371-379 over biquad.rs:263-327) as a
// systolic array across lanes.
//
// Inside the token-ring chain kernel the EQ is two serial units of five sections; with the suppressor on, that kernel is
// what the step waits for (it cannot use more than one CU per 64 streams), while the suppressor's CUs have vector issue
// to spare.  So in that pipeline the EQ runs here, on the suppressor's side, and the chain launch of the window skips its
// EQ units and its input unit (measured with the stage subsets of tools/bench_chain_stages.py: 39.2 -> 33.5 ms per 2 s).
//
// Layout: a stream is a row of 16 lanes, lane k = biquad section k (a wave holds four streams; configurations with more
// than 16 sections keep the EQ inside the chain kernel).  At step T lane k filters sample T - k: its input is what lane
// k - 1 produced one step earlier (DPP row_shr:1, register to register), lane 0 takes the next input sample.  Lanes past
// the last section hand their input on unchanged, so the filtered sample always leaves from lane 15, fifteen steps after
// it entered.  A section's own memories never leave its lane's registers.  Per stream this is exactly the reference's
// sample order through exactly the reference's operations (f64 direct form II transposed, the result rounded to f32
// between sections): the same bits as the chain kernel's EQ units.
//
// Input and output move in groups of 16 samples per row: lane j of a row loads sample 16 m + j (a 64-byte run per stream),
// lane 0 picks sample j of the group with DPP row_shl:j, and the sample leaving lane 15 is dropped into lane j of the
// output register with row_shl:(15 - j); loads run two groups ahead of their use.  Lane 0 also keeps the block input
// statistics (sum of squares in f64 in sample order, peak) the chain kernel's input unit would have produced.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "af_dsp.h"
#include "af_eq_systolic_body.h"

namespace af {

template <bool kStats, bool kXf, bool kPower = false>
__global__ __launch_bounds__(64) void eq_systolic_kernel(EqSystolicArgs a) {
  eq_systolic_body<kStats, kXf, kPower>(a, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same EQ with a LANE per stream (round 3): one wave takes 64 streams through every section, memories and coefficients in
// registers, four samples per round written out so that the scheduler sees the (sample, section) grid and walks it along its
// diagonals -- (n, k) needs (n, k-1)'s output and (n-1, k)'s memories, so ten sections give ten independent chains.  Per
// stream and sample this is 11 instructions per section where the systolic form spends ~23 wave instructions per step of FOUR
// streams (every lane of a 16-lane row executes the step, six of them idle for ten sections): a third of the issue slots,
// on the CUs the suppressor's kernels need.  What it gives up is latency hiding by occupancy: 64 waves per 4096 streams, each
// wanting a SIMD's whole issue rate (priority 2).  Same expressions in the same order as the systolic body and the chain
// kernel's EQ units: the same bits.  Serves windows without a pending crossfade, one preset, stream-major audio with 16-byte
// rows; everything else takes the systolic kernel.
// `sec0`: the first of this launch's kSec sections; `head`: the launch takes the raw input (scrub / clamp apply; kStats only there).
// A window's EQ may run as TWO launches, sections [0, h) and [h, n), on two streams: the sample between two sections is an f32
// in the reference, so the hand-over through the audio buffer is exact, and the second half of window w runs beside the first
// half of window w + 1 -- the EQ's period per window halves (one launch per window was the pipeline's longest stage).
// Four waves (four 64-stream groups) per workgroup, one per SIMD of a CU: a wave of this kernel holds ~240 vector registers -- half
// a SIMD's file -- for milliseconds; one such wave per CU on 64 CUs takes two of three wave slots away from the 169-register
// transform kernels on a third of the suppressor's CUs (their workgroups need a slot on every SIMD).  Packed, sixteen CUs carry them.
constexpr int kEqStreamWaves = 4;
template <int kSec, bool kStats, bool kPower>
__global__ __launch_bounds__(64 * kEqStreamWaves) void eq_stream_kernel(EqSystolicArgs a, int sec0, int head) {
  const int lane = threadIdx.x & 63;
  const int group = blockIdx.x * kEqStreamWaves + (threadIdx.x >> 6);
  if (group * 64 >= a.n_streams) return;  // (no workgroup barrier in this kernel)
  const int s = group * 64 + lane;
  const bool valid = s < a.n_streams;
  const int sc = valid ? s : a.n_streams - 1;
  const int64_t NS = a.n_streams;
  const ChainParams &P = a.params[0];
  const uint32_t flags = P.flags;
  const bool scrub = head && (flags & (kFlagInputScrub | kFlagInputClamp)) != 0, clamp = head && (flags & kFlagInputClamp) != 0;
  const int cb = P.control_block;
#ifndef AF_EQ_STREAM_NO_PRIO
  __builtin_amdgcn_s_setprio(2);
#endif
  BiquadCoef c[kSec];
  double z1[kSec], z2[kSec];
#pragma unroll
  for (int k = 0; k < kSec; ++k) {
    const SectionParams &sp = P.eq[sec0 + k];
    c[k] = sp.xf_remaining > 0 ? sp.pending : sp.active;  // (a crossfade that ended in an earlier launch: see the systolic body)
    z1[k] = a.st64[(int64_t)(kEqBase + 4 * (sec0 + k)) * NS + sc];
    z2[k] = a.st64[(int64_t)(kEqBase + 4 * (sec0 + k) + 1) * NS + sc];
  }
  const float *row_in = a.in + (int64_t)sc * a.stream_stride;
  float *row_out = a.audio + (int64_t)sc * a.stream_stride;
  const int64_t n = a.n_samples;
  double in_sq = 0.0, out_sq = 0.0;
  float in_peak = 0.0f;
  int64_t block_index = 0;
  int in_block = 0;  // samples of the current control block done (wave-uniform)
  auto sample = [&](float v) -> float {
    if (scrub && !finite_f32(v)) v = 0.0f;
    if (clamp) v = fclamp(v, -1.0f, 1.0f);
    if (kStats) {
      const double xd0 = (double)v;
      in_sq += xd0 * xd0;
      in_peak = fmaxf(in_peak, fabsf(v));
    }
    float x = v;
#pragma unroll
    for (int k = 0; k < kSec; ++k) {
      const double xd = (double)x;
      const double y = c[k].b0 * xd + z1[k];
      z1[k] = c[k].b1 * xd - c[k].a1 * y + z2[k];
      z2[k] = c[k].b2 * xd - c[k].a2 * y;
      x = (float)y;
    }
    if (kPower) {
      const double yd = (double)x;
      out_sq += finite_f32(x) ? yd * yd : 0.0;
    }
    return x;
  };
  auto flush = [&]() {  // a control block (or the launch) has ended
    if (kStats && valid && a.stats) {
      BlockStats &r = a.stats[block_index * NS + s];
      r.input_square_sum = in_sq;
      r.input_sample_peak = in_peak;
    }
    if (kPower && valid) a.block_power[block_index * NS + s] = out_sq;
    block_index += 1;
    in_block = 0;
    in_sq = 0.0;
    in_peak = 0.0f;
    out_sq = 0.0;
  };
  const int64_t quads = n >> 2;
  // the loads run two rounds ahead of the arithmetic (a lane walks its own row: a 128-byte line serves eight rounds)
  float4 v0 = quads > 0 ? *reinterpret_cast<const float4 *>(row_in) : make_float4(0, 0, 0, 0);
  float4 v1 = quads > 1 ? *reinterpret_cast<const float4 *>(row_in + 4) : make_float4(0, 0, 0, 0);
  for (int64_t q = 0; q < quads; ++q) {
    const float4 v2 = q + 2 < quads ? *reinterpret_cast<const float4 *>(row_in + 4 * (q + 2)) : make_float4(0, 0, 0, 0);
    float4 o;
    if (in_block + 4 <= cb) {  // (wave-uniform) the plain round: no block boundary inside
      o.x = sample(v0.x);
      o.y = sample(v0.y);
      o.z = sample(v0.z);
      o.w = sample(v0.w);
      in_block += 4;
      if (in_block == cb) flush();
    } else {  // a control block that is not a multiple of four samples long ends inside the round
      float t[4] = {v0.x, v0.y, v0.z, v0.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[j] = sample(t[j]);
        if (++in_block == cb) flush();
      }
      o = make_float4(t[0], t[1], t[2], t[3]);
    }
    if (valid) *reinterpret_cast<float4 *>(row_out + 4 * q) = o;
    v0 = v1;
    v1 = v2;
  }
  for (int64_t t = quads * 4; t < n; ++t) {  // ragged end
    const float o = sample(row_in[t]);
    if (valid) row_out[t] = o;
    if (++in_block == cb) flush();
  }
  if (in_block > 0) flush();  // a short last block
  if (valid) {
#pragma unroll
    for (int k = 0; k < kSec; ++k) {
      a.st64[(int64_t)(kEqBase + 4 * (sec0 + k)) * NS + s] = z1[k];
      a.st64[(int64_t)(kEqBase + 4 * (sec0 + k) + 1) * NS + s] = z2[k];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The lane-per-stream EQ with its sections split over TWO waves per 64-stream group: the head wave runs sections [0, kSecA), drops
// its output -- an f32 in the reference as well -- into an LDS ring, the tail wave takes it through [kSecA, kSecA + kSecB) and
// stores the audio.  Same arithmetic, half the dependent work per wave: a window's EQ takes half as long (as one wave per group it
// had become the pipeline's longest stage: 52 windows x 3.3 ms = the step).  Four groups = eight waves per workgroup, two per
// SIMD.  Ring: kRingBlocks blocks of 16 samples per group; `produced` / `consumed` count blocks (workgroup-scope release / acquire,
// one writer each).  Bounded polls: a partner that never arrives ends the wait after ~1 s and the kernel runs out (garbage audio,
// never expected: both waves of a group are of the same workgroup and resident together).  Serves sample counts that are multiples
// of four; anything else takes the one-wave kernel.
constexpr int kEq2Groups = 2, kEq2RingBlocks = 4, kEq2BlockQuads = 4;  // (two groups = four waves: one per SIMD -- with four groups two waves share a SIMD and each runs at half speed: nothing gained)
struct Eq2Lds {
  float4 ring[kEq2Groups][kEq2RingBlocks][kEq2BlockQuads][64];
  int produced[kEq2Groups], consumed[kEq2Groups];
};
template <int kSecA, int kSecB, bool kStats, bool kPower>
__global__ __launch_bounds__(128 * kEq2Groups) void eq_stream2_kernel(EqSystolicArgs a) {
  __shared__ Eq2Lds L;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = wave >> 1;
  const bool head = (wave & 1) == 0;
  if (threadIdx.x < kEq2Groups) { L.produced[threadIdx.x] = 0; L.consumed[threadIdx.x] = 0; }
  __syncthreads();
  const int group = blockIdx.x * kEq2Groups + g;
  if (group * 64 >= a.n_streams) return;  // (both waves of the group leave: no barrier after this point)
  const int s = group * 64 + lane;
  const bool valid = s < a.n_streams;
  const int sc = valid ? s : a.n_streams - 1;
  const int64_t NS = a.n_streams;
  const ChainParams &P = a.params[0];
  const uint32_t flags = P.flags;
  const int cb = P.control_block;
  const int64_t quads = a.n_samples >> 2;
  const int64_t blocks = (quads + kEq2BlockQuads - 1) / kEq2BlockQuads;
  __builtin_amdgcn_s_setprio(2);
  auto wait_for = [&](int *word, int64_t at_least) {
    int spins = 0;
    while ((int64_t)__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < at_least) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 24)) break;
    }
  };
  if (head) {
    constexpr int kSec = kSecA;
    const bool scrub = (flags & (kFlagInputScrub | kFlagInputClamp)) != 0, clamp = (flags & kFlagInputClamp) != 0;
    BiquadCoef c[kSec];
    double z1[kSec], z2[kSec];
#pragma unroll
    for (int k = 0; k < kSec; ++k) {
      const SectionParams &sp = P.eq[k];
      c[k] = sp.xf_remaining > 0 ? sp.pending : sp.active;
      z1[k] = a.st64[(int64_t)(kEqBase + 4 * k) * NS + sc];
      z2[k] = a.st64[(int64_t)(kEqBase + 4 * k + 1) * NS + sc];
    }
    const float *row_in = a.in + (int64_t)sc * a.stream_stride;
    double in_sq = 0.0;
    float in_peak = 0.0f;
    int64_t block_index = 0;
    int in_block = 0;
    auto sample = [&](float v) -> float {
      if (scrub && !finite_f32(v)) v = 0.0f;
      if (clamp) v = fclamp(v, -1.0f, 1.0f);
      if (kStats) {
        const double xd0 = (double)v;
        in_sq += xd0 * xd0;
        in_peak = fmaxf(in_peak, fabsf(v));
      }
      float x = v;
#pragma unroll
      for (int k = 0; k < kSec; ++k) {
        const double xd = (double)x;
        const double y = c[k].b0 * xd + z1[k];
        z1[k] = c[k].b1 * xd - c[k].a1 * y + z2[k];
        z2[k] = c[k].b2 * xd - c[k].a2 * y;
        x = (float)y;
      }
      return x;
    };
    auto flush = [&]() {
      if (kStats && valid && a.stats) {
        BlockStats &r = a.stats[block_index * NS + s];
        r.input_square_sum = in_sq;
        r.input_sample_peak = in_peak;
      }
      block_index += 1;
      in_block = 0;
      in_sq = 0.0;
      in_peak = 0.0f;
    };
    float4 v0 = quads > 0 ? *reinterpret_cast<const float4 *>(row_in) : make_float4(0, 0, 0, 0);
    float4 v1 = quads > 1 ? *reinterpret_cast<const float4 *>(row_in + 4) : make_float4(0, 0, 0, 0);
    for (int64_t b = 0; b < blocks; ++b) {
      if (b >= kEq2RingBlocks) wait_for(&L.consumed[g], b - kEq2RingBlocks + 1);  // the slot has been read
      float4(&slot)[kEq2BlockQuads][64] = L.ring[g][b % kEq2RingBlocks];
#pragma unroll
      for (int j = 0; j < kEq2BlockQuads; ++j) {
        const int64_t q = b * kEq2BlockQuads + j;
        if (q < quads) {
          const float4 v2 = q + 2 < quads ? *reinterpret_cast<const float4 *>(row_in + 4 * (q + 2)) : make_float4(0, 0, 0, 0);
          float t[4] = {v0.x, v0.y, v0.z, v0.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            t[u] = sample(t[u]);
            if (++in_block == cb) flush();
          }
          slot[j][lane] = make_float4(t[0], t[1], t[2], t[3]);
          v0 = v1;
          v1 = v2;
        }
      }
      __hip_atomic_store(&L.produced[g], (int)(b + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (in_block > 0) flush();
    if (valid) {
#pragma unroll
      for (int k = 0; k < kSec; ++k) {
        a.st64[(int64_t)(kEqBase + 4 * k) * NS + s] = z1[k];
        a.st64[(int64_t)(kEqBase + 4 * k + 1) * NS + s] = z2[k];
      }
    }
  } else {
    constexpr int kSec = kSecB;
    BiquadCoef c[kSec];
    double z1[kSec], z2[kSec];
#pragma unroll
    for (int k = 0; k < kSec; ++k) {
      const SectionParams &sp = P.eq[kSecA + k];
      c[k] = sp.xf_remaining > 0 ? sp.pending : sp.active;
      z1[k] = a.st64[(int64_t)(kEqBase + 4 * (kSecA + k)) * NS + sc];
      z2[k] = a.st64[(int64_t)(kEqBase + 4 * (kSecA + k) + 1) * NS + sc];
    }
    float *row_out = a.audio + (int64_t)sc * a.stream_stride;
    double out_sq = 0.0;
    int64_t block_index = 0;
    int in_block = 0;
    auto sample = [&](float x) -> float {
#pragma unroll
      for (int k = 0; k < kSec; ++k) {
        const double xd = (double)x;
        const double y = c[k].b0 * xd + z1[k];
        z1[k] = c[k].b1 * xd - c[k].a1 * y + z2[k];
        z2[k] = c[k].b2 * xd - c[k].a2 * y;
        x = (float)y;
      }
      if (kPower) {
        const double yd = (double)x;
        out_sq += finite_f32(x) ? yd * yd : 0.0;
      }
      return x;
    };
    auto flush = [&]() {
      if (kPower && valid) a.block_power[block_index * NS + s] = out_sq;
      block_index += 1;
      in_block = 0;
      out_sq = 0.0;
    };
    for (int64_t b = 0; b < blocks; ++b) {
      wait_for(&L.produced[g], b + 1);
      float4(&slot)[kEq2BlockQuads][64] = L.ring[g][b % kEq2RingBlocks];
#pragma unroll
      for (int j = 0; j < kEq2BlockQuads; ++j) {
        const int64_t q = b * kEq2BlockQuads + j;
        if (q < quads) {
          const float4 v = slot[j][lane];
          float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            t[u] = sample(t[u]);
            if (++in_block == cb) flush();
          }
          if (valid) *reinterpret_cast<float4 *>(row_out + 4 * q) = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
      __hip_atomic_store(&L.consumed[g], (int)(b + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (in_block > 0) flush();
    if (valid) {
#pragma unroll
      for (int k = 0; k < kSec; ++k) {
        a.st64[(int64_t)(kEqBase + 4 * (kSecA + k)) * NS + s] = z1[k];
        a.st64[(int64_t)(kEqBase + 4 * (kSecA + k) + 1) * NS + s] = z2[k];
      }
    }
  }
}
template <int kSecA, int kSecB>
static void launch_eq_stream2_sections(const EqSystolicArgs &a, bool stats, bool power, hipStream_t stream) {
  const unsigned groups = (unsigned)((a.n_streams + 63) / 64);
  const dim3 grid((groups + kEq2Groups - 1) / kEq2Groups), block(128 * kEq2Groups);
  if (power) hipLaunchKernelGGL((eq_stream2_kernel<kSecA, kSecB, true, true>), grid, block, 0, stream, a);
  else if (stats) hipLaunchKernelGGL((eq_stream2_kernel<kSecA, kSecB, true, false>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((eq_stream2_kernel<kSecA, kSecB, false, false>), grid, block, 0, stream, a);
}
static bool launch_eq_stream2(const EqSystolicArgs &a, int n_sections, bool stats, bool power, hipStream_t stream) {
  if ((a.n_samples & 3) != 0) return false;
  switch (n_sections) {
#define AF_EQ_STREAM2_CASE(n) case n: launch_eq_stream2_sections<(n) / 2, (n) - (n) / 2>(a, stats, power, stream); return true;
    AF_EQ_STREAM2_CASE(2) AF_EQ_STREAM2_CASE(3) AF_EQ_STREAM2_CASE(4) AF_EQ_STREAM2_CASE(5) AF_EQ_STREAM2_CASE(6) AF_EQ_STREAM2_CASE(7)
    AF_EQ_STREAM2_CASE(8) AF_EQ_STREAM2_CASE(9) AF_EQ_STREAM2_CASE(10) AF_EQ_STREAM2_CASE(11) AF_EQ_STREAM2_CASE(12) AF_EQ_STREAM2_CASE(13)
    AF_EQ_STREAM2_CASE(14) AF_EQ_STREAM2_CASE(15) AF_EQ_STREAM2_CASE(16)
#undef AF_EQ_STREAM2_CASE
    default: return false;
  }
}

template <int kSec>
static void launch_eq_stream_sections(const EqSystolicArgs &a, bool stats, bool power, int sec0, bool head, hipStream_t stream) {
  const unsigned groups = (unsigned)((a.n_streams + 63) / 64);
  const dim3 grid((groups + kEqStreamWaves - 1) / kEqStreamWaves), block(64 * kEqStreamWaves);
  const int h = head ? 1 : 0;
  if (power && stats) hipLaunchKernelGGL((eq_stream_kernel<kSec, true, true>), grid, block, 0, stream, a, sec0, h);
  else if (power) hipLaunchKernelGGL((eq_stream_kernel<kSec, false, true>), grid, block, 0, stream, a, sec0, h);
  else if (stats) hipLaunchKernelGGL((eq_stream_kernel<kSec, true, false>), grid, block, 0, stream, a, sec0, h);
  else hipLaunchKernelGGL((eq_stream_kernel<kSec, false, false>), grid, block, 0, stream, a, sec0, h);
}
static bool launch_eq_stream(const EqSystolicArgs &a, int n_sections, bool stats, bool power, hipStream_t stream, int sec0 = 0,
                             bool head = true) {
  switch (n_sections) {
#define AF_EQ_STREAM_CASE(k) case k: launch_eq_stream_sections<k>(a, stats, power, sec0, head, stream); return true;
    AF_EQ_STREAM_CASE(1) AF_EQ_STREAM_CASE(2) AF_EQ_STREAM_CASE(3) AF_EQ_STREAM_CASE(4) AF_EQ_STREAM_CASE(5) AF_EQ_STREAM_CASE(6)
    AF_EQ_STREAM_CASE(7) AF_EQ_STREAM_CASE(8) AF_EQ_STREAM_CASE(9) AF_EQ_STREAM_CASE(10) AF_EQ_STREAM_CASE(11) AF_EQ_STREAM_CASE(12)
    AF_EQ_STREAM_CASE(13) AF_EQ_STREAM_CASE(14) AF_EQ_STREAM_CASE(15) AF_EQ_STREAM_CASE(16)
#undef AF_EQ_STREAM_CASE
    default: return false;
  }
}

// One half of a window's EQ in the lane-per-stream form: sections [sec0, sec0 + count) of the single preset.  `head`: this launch
// takes the raw input (scrub / clamp, block input statistics into `stats`); a launch that ends at the last section may keep the
// block powers (`block_power`).  Returns hipErrorNotSupported when the form does not serve the buffers (the caller then runs the
// whole EQ through launch_eq_systolic).
hipError_t launch_eq_stream_part(const ChainParams *d_params, double *st64, const float *in, float *audio, BlockStats *stats,
                                 double *block_power, int sec0, int count, bool head, int64_t n_samples, int64_t stream_stride,
                                 int32_t n_streams, hipStream_t stream) {
  if (count <= 0 || count > 16 || (stream_stride % 4) != 0 || !audio ||
      ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(audio)) & 15) != 0)
    return hipErrorNotSupported;
  EqSystolicArgs a{d_params, nullptr, st64, in, audio, nullptr, nullptr, head ? stats : nullptr, n_samples, stream_stride, 0, n_streams, 0,
                   block_power};
  // (block powers are kept together with block rows: the kernel's flush writes both; without `stats` the rows are skipped)
  if (!launch_eq_stream(a, count, head && stats != nullptr, block_power != nullptr, stream, sec0, head)) return hipErrorNotSupported;
  return hipGetLastError();
}

// `audio`: stream-major output (may be `in`); or null and `ring` / `ring_in` / `ring_rows` / `n0`: the stage pipeline's rings.
// `stats` null: no block input statistics.  `crossfade`: some section has a coefficient crossfade pending.
// `block_power` ([block][stream], with `stats`): also the square sum of every control block of the filtered samples -- the
// launch is then the pre-pass of an auto-makeup window.
hipError_t launch_eq_systolic(const ChainParams *d_params, const int32_t *d_group_preset, double *st64, const float *in, float *audio,
                              float *ring, float *ring_in, int32_t ring_rows, int64_t n0, BlockStats *stats, bool crossfade,
                              int64_t n_samples, int64_t stream_stride, int32_t n_streams, hipStream_t stream, double *block_power,
                              int n_sections) {
  EqSystolicArgs a{d_params, d_group_preset, st64, in, audio, ring, ring_in, stats, n_samples, stream_stride, n0, n_streams, ring_rows,
                   block_power};
  // `n_sections` > 0: the caller knows the (single) preset runs that many EQ sections -- the lane-per-stream kernel where it
  // serves the launch (AF_EQ_STREAM=0: always the systolic kernel)
  static const bool stream_form = [] {
    const char *env = std::getenv("AF_EQ_STREAM");
    return !env || std::atoi(env) != 0;
  }();
  if (stream_form && n_sections > 0 && !crossfade && !d_group_preset && audio && !ring && (stream_stride % 4) == 0 &&
      ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(audio)) & 15) == 0 && (!block_power || stats)) {
    static const bool two_waves = [] {  // AF_EQ_STREAM=1: one wave per group (all sections); default: two waves, sections split
      const char *env = std::getenv("AF_EQ_STREAM");
      return !env || std::atoi(env) != 1;
    }();
    if (two_waves && launch_eq_stream2(a, n_sections, stats != nullptr, block_power != nullptr, stream)) return hipGetLastError();
    if (launch_eq_stream(a, n_sections, stats != nullptr, block_power != nullptr, stream)) return hipGetLastError();
  }
  const dim3 grid((unsigned)((n_streams + 3) / 4)), block(64);
  if (block_power) {
    if (!stats) return hipErrorInvalidValue;
    if (crossfade) hipLaunchKernelGGL((eq_systolic_kernel<true, true, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((eq_systolic_kernel<true, false, true>), grid, block, 0, stream, a);
  } else if (crossfade) {
    if (stats) hipLaunchKernelGGL((eq_systolic_kernel<true, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((eq_systolic_kernel<false, true>), grid, block, 0, stream, a);
  } else {
    if (stats) hipLaunchKernelGGL((eq_systolic_kernel<true, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((eq_systolic_kernel<false, false>), grid, block, 0, stream, a);
  }
  return hipGetLastError();
}

}  // namespace af
