// af_eq_systolic_body.h -- the systolic EQ as a device function of (arguments, index of a 64-thread block): launched as a
// kernel of its own (af_eq_systolic.hip) and as one role of the stage pipeline's launches (af_stages.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "af_dsp.h"

namespace af {

struct EqSystolicArgs {
  const ChainParams *params;    // device; an array when `group_preset` is set
  const int32_t *group_preset;  // [ceil(n_streams / 64)] or null
  double *st64;                 // [f64 fields][n_streams]: the sections' memories (kEqBase + 4 sec + {0, 1}; + {2, 3} of a pending filter)
  const float *in;              // [stream][stride]
  float *audio;                 // [stream][stride]: the filtered samples (may be `in`: reads run three groups ahead of the stores); or null:
  float *ring;                  // the filtered samples go into the stage pipeline's ring (af_stages.h), sample t at absolute n0 + t
  float *ring_in;               // with `ring`: the scrubbed / clamped input goes here (the input-statistics stage reads it)
  BlockStats *stats;            // [block][stream]: input_square_sum / input_sample_peak are written here (kStats)
  int64_t n_samples, stream_stride, n0;
  int32_t n_streams, ring_rows;
  double *block_power;          // [block][stream]: sum of squares of the filtered samples of each control block (kPower), or null
  const float *ring_src;        // the input comes from this ring of the stage pipeline (the de-esser stages' output) instead of `in`
};

template <int N>
__device__ __forceinline__ float row_shl(float v) {  // lane i of a 16-lane row receives lane i + N (N = 0: itself)
  if constexpr (N == 0) return v;
  else return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x100 + N, 0xf, 0xf, false));  // (lanes past the row's end keep their own value: never read)
}
// lane i receives `v` of lane i - 1; lane 0 of a row has no left neighbour and keeps `keep`
__device__ __forceinline__ float row_shr1_keep(float keep, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(v), 0x111, 0xf, 0xf, false));
}

// kStats: lane 0 keeps the block input statistics.  kXf: a coefficient crossfade (biquad.rs:263-327) is pending for some
// section at the start of the launch: every lane carries its pending filter beside the active one (general, slower form;
// a stream opens with at most a few hundred such samples).  kPower: lane 15 keeps the square sum of every control block of
// the FILTERED samples (f64, sample order, non-finite samples skipped: what the token-ring kernel's pre-pass launch leaves
// in `output_square_sum`) -- the compressor-input block power the auto-makeup controller needs before the block's first
// sample (compressor.rs:583-596,710), so this kernel is the whole pre-pass of an auto-makeup window.
template <bool kStats, bool kXf, bool kPower = false>
__device__ __forceinline__ void eq_systolic_body(const EqSystolicArgs &a, int block) {
  const int lane = threadIdx.x & 63;
  const int k = lane & 15;                      // section
  const int s = block * 4 + (lane >> 4);        // stream
  const bool valid = s < a.n_streams;
  const int sc = valid ? s : a.n_streams - 1;
  const int64_t NS = a.n_streams;
  const ChainParams &P = a.params[a.group_preset ? a.group_preset[sc >> 6] : 0];
  const int nsec = (P.flags & kFlagEq) ? P.n_eq_sections : 0;
  const bool sec_lane = k < nsec;
  const uint32_t flags = P.flags;
  const bool scrub = (flags & (kFlagInputScrub | kFlagInputClamp)) != 0, clamp = (flags & kFlagInputClamp) != 0;
  const int cb = P.control_block;

  // this lane's section: coefficients (a crossfade that ended in an earlier launch has left its target coefficients in
  // `pending` until the host promotes them) and memories
  BiquadCoef c{1.0, 0.0, 0.0, 0.0, 0.0}, p{1.0, 0.0, 0.0, 0.0, 0.0};
  double z1 = 0.0, z2 = 0.0, pz1 = 0.0, pz2 = 0.0;
  int rem = 0;
  double xf_total = 1.0;
  int xf_total_i = 0;
  if (sec_lane) {
    const SectionParams &sp = P.eq[k];
    z1 = a.st64[(int64_t)(kEqBase + 4 * k) * NS + sc];
    z2 = a.st64[(int64_t)(kEqBase + 4 * k + 1) * NS + sc];
    if (kXf) {
      c = sp.active;
      p = sp.pending;
      rem = sp.xf_remaining;
      xf_total_i = sp.xf_total;
      xf_total = (double)sp.xf_total;
      pz1 = a.st64[(int64_t)(kEqBase + 4 * k + 2) * NS + sc];
      pz2 = a.st64[(int64_t)(kEqBase + 4 * k + 3) * NS + sc];
    } else {
      c = sp.xf_remaining > 0 ? sp.pending : sp.active;
    }
  }

  float *row = a.audio ? a.audio + (int64_t)sc * a.stream_stride : nullptr;
  float *ring = a.ring ? a.ring + (int64_t)(sc >> 6) * a.ring_rows * kLanes : nullptr;
  float *ring_in = a.ring_in ? a.ring_in + (int64_t)(sc >> 6) * a.ring_rows * kLanes : nullptr;
  const int ring_lane = sc & 63;
  auto ring_at = [&](int64_t t) {  // a lane's four consecutive samples are contiguous in the rings (af_stages.h)
    const int64_t na = a.n0 + t;
    return ((na >> 2) & (int64_t)(a.ring_rows / 4 - 1)) * (kLanes * 4) + ring_lane * 4 + (na & 3);
  };
  const float *row_in = a.in + (int64_t)sc * a.stream_stride;
  const float *ring_src = a.ring_src ? a.ring_src + (int64_t)(sc >> 6) * a.ring_rows * kLanes : nullptr;
  const int64_t n = a.n_samples;
  const int64_t groups = (n + 15) / 16;
  auto fetch = [&](int64_t g) -> float {
    const int64_t t = g * 16 + k;
    float v = (g < groups && t < n) ? (ring_src ? ring_src[ring_at(t)] : row_in[t]) : 0.0f;
    if (scrub && !finite_f32(v)) v = 0.0f;  // python_api.rs:515-523 / routing.rs:802-823 (every lane scrubs its own sample)
    if (clamp) v = fclamp(v, -1.0f, 1.0f);
    if (ring_in && valid && g < groups && t < n) ring_in[ring_at(t)] = v;
    return v;
  };
  auto put = [&](int64_t t, float v) {  // filtered sample t of this lane's stream
    if (row) row[t] = v;
    else ring[ring_at(t)] = v;
  };
  float x_cur = fetch(0), x_n1 = fetch(1), x_n2 = fetch(2);
  float y_prev = 0.0f;   // what this lane produced at the previous step
  float out_reg = 0.0f;  // lane j: output sample 16 m + j of the group being collected
  double in_sq = 0.0;    // block input statistics (block_processor.rs:111-118); only lane 0's are used
  float in_peak = 0.0f;
  int64_t block_index = 0;
  int in_block = 0;      // samples of the current control block consumed so far (wave-uniform: streams advance in lock step)
  double out_sq = 0.0;   // kPower: square sum of the current control block's filtered samples; only lane 15's is used
  int64_t out_block_index = 0;
  int out_in_block = 0;  // filtered samples of the current control block that have left lane 15 (wave-uniform)

  // Steps T = 16 g + j.  The sample leaving lane 15 at step T is sample u = T - 15 = 16 (g - 1) + j + 1: position j + 1 of
  // output group g - 1 for j < 15, position 0 of group g for j = 15.  So group G is complete after step (G + 1, 14) and is
  // stored at step (G + 1, 15), just before lane 0 takes the first sample of group G + 1; reads run three groups ahead of
  // the stores (the buffer may be filtered in place).
  //
  // Three forms of a group: kEdge = some lane's sample index is still negative or already past the end (first and last
  // groups); kFlush = a control block may end inside the group (checked per step); the plain form is branch-free.  The
  // square and the peak are accumulated by every lane (only lane 0's are ever used).  Lanes past the last section run an
  // identity filter and select their input.
  auto group = [&](int64_t g, auto edge_tag, auto flush_tag) {
    constexpr bool kEdge = decltype(edge_tag)::value, kFlush = decltype(flush_tag)::value;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int64_t T = g * 16 + j;
      // ---- input of this step: lane 0 <- lane j of the group register, the others <- their left neighbour's last result
      float xin;
      switch (j) {
        case 0: xin = row_shl<0>(x_cur); break;
        case 1: xin = row_shl<1>(x_cur); break;
        case 2: xin = row_shl<2>(x_cur); break;
        case 3: xin = row_shl<3>(x_cur); break;
        case 4: xin = row_shl<4>(x_cur); break;
        case 5: xin = row_shl<5>(x_cur); break;
        case 6: xin = row_shl<6>(x_cur); break;
        case 7: xin = row_shl<7>(x_cur); break;
        case 8: xin = row_shl<8>(x_cur); break;
        case 9: xin = row_shl<9>(x_cur); break;
        case 10: xin = row_shl<10>(x_cur); break;
        case 11: xin = row_shl<11>(x_cur); break;
        case 12: xin = row_shl<12>(x_cur); break;
        case 13: xin = row_shl<13>(x_cur); break;
        case 14: xin = row_shl<14>(x_cur); break;
        default: xin = row_shl<15>(x_cur); break;
      }
      const float in = row_shr1_keep(xin, y_prev);  // (lane 0 keeps the group register's sample)
      const double xd = (double)in;
      if (kStats && (!kEdge || T < n)) {  // lane 0 sees sample T
        in_sq += xd * xd;
        in_peak = fmaxf(in_peak, fabsf(in));
        if (kFlush) {
          in_block += 1;
          if (in_block == cb || T + 1 == n) {  // wave-uniform
            if (k == 0 && valid && a.stats) {
              BlockStats &r = a.stats[block_index * NS + s];
              r.input_square_sum = in_sq;
              r.input_sample_peak = in_peak;
            }
            block_index += 1;
            in_block = 0;
            in_sq = 0.0;
            in_peak = 0.0f;
          }
        }
      }
      double y = c.b0 * xd + z1;
      const double nz1 = c.b1 * xd - c.a1 * y + z2;
      const double nz2 = c.b2 * xd - c.a2 * y;
      if (kXf) {
        const int64_t t = T - k;  // the sample this lane sees at this step
        if (t >= 0 && t < n) {
          z1 = nz1;
          z2 = nz2;
          if (rem > 0) {
            const double yp = p.b0 * xd + pz1;
            pz1 = p.b1 * xd - p.a1 * yp + pz2;
            pz2 = p.b2 * xd - p.a2 * yp;
            const double fade = (double)(xf_total_i - rem + 1) / xf_total;
            y = y * (1.0 - fade) + yp * fade;
            rem -= 1;
            if (rem == 0) {  // promote_pending_coefficients, biquad.rs:276-286
              c = p;
              z1 = pz1;
              z2 = pz2;
            }
          }
        }
      } else if (kEdge) {
        const int64_t t = T - k;  // the sample this lane sees at this step
        if (t >= 0 && t < n) {
          z1 = nz1;
          z2 = nz2;
        }
      } else {
        z1 = nz1;
        z2 = nz2;
      }
      y_prev = sec_lane ? (float)y : in;  // lanes past the last section pass their input on
      if (kPower && (!kEdge || (T >= 15 && T - 15 < n))) {  // lane 15 has just produced filtered sample T - 15
        const double yd = (double)y_prev;
        out_sq += finite_f32(y_prev) ? yd * yd : 0.0;
        if (kFlush) {
          out_in_block += 1;
          if (out_in_block == cb || T - 15 + 1 == n) {  // wave-uniform
            if (k == 15 && valid) a.block_power[out_block_index * NS + s] = out_sq;
            out_block_index += 1;
            out_in_block = 0;
            out_sq = 0.0;
          }
        }
      }
      // ---- the sample leaving lane 15 now is sample T - 15 = position (j + 1) & 15 of its output group
      float leaving;
      switch (j) {  // lane (j + 1) & 15 <- lane 15: row_shl by 15 - ((j + 1) & 15)
        case 0: leaving = row_shl<14>(y_prev); break;
        case 1: leaving = row_shl<13>(y_prev); break;
        case 2: leaving = row_shl<12>(y_prev); break;
        case 3: leaving = row_shl<11>(y_prev); break;
        case 4: leaving = row_shl<10>(y_prev); break;
        case 5: leaving = row_shl<9>(y_prev); break;
        case 6: leaving = row_shl<8>(y_prev); break;
        case 7: leaving = row_shl<7>(y_prev); break;
        case 8: leaving = row_shl<6>(y_prev); break;
        case 9: leaving = row_shl<5>(y_prev); break;
        case 10: leaving = row_shl<4>(y_prev); break;
        case 11: leaving = row_shl<3>(y_prev); break;
        case 12: leaving = row_shl<2>(y_prev); break;
        case 13: leaving = row_shl<1>(y_prev); break;
        case 14: leaving = row_shl<0>(y_prev); break;
        default: leaving = row_shl<15>(y_prev); break;  // j = 15: sample 16 g, lane 0 of the NEXT group
      }
      if (j == 15) {
        // group g - 1 is complete (its sample 0 was kept in lane 0 since the previous group's last step): store it, then
        // start group g with its first sample
        const int64_t t_out = (g - 1) * 16 + k;
        if (g >= 1 && valid && t_out < n) put(t_out, out_reg);
        if (k == 0) out_reg = leaving;
      } else if (k == ((j + 1) & 15)) {
        out_reg = leaving;
      }
    }
    if (kStats && !kFlush) in_block += 16;
    if (kPower && !kFlush) out_in_block += 16;
    x_cur = x_n1;
    x_n1 = x_n2;
    x_n2 = fetch(g + 3);
  };
  for (int64_t g = 0; g < groups + 1; ++g) {
    const int left = __builtin_amdgcn_readfirstlane(cb - in_block);  // samples to the end of the control block
    const int left_out = kPower ? __builtin_amdgcn_readfirstlane(cb - out_in_block) : 32;  // ... on lane 15's side
    if (kXf || g < 1 || (g + 1) * 16 >= n) group(g, std::true_type{}, std::true_type{});  // (>=: a short last block ends at T + 1 == n)
    else if ((kStats && left <= 16) || (kPower && left_out <= 16)) group(g, std::false_type{}, std::true_type{});
    else group(g, std::false_type{}, std::false_type{});
  }
  // (group `groups - 1`, the last one, was stored by step 15 of the extra iteration g = groups)
  if (sec_lane && valid) {
    a.st64[(int64_t)(kEqBase + 4 * k) * NS + s] = z1;
    a.st64[(int64_t)(kEqBase + 4 * k + 1) * NS + s] = z2;
    if (kXf) {
      a.st64[(int64_t)(kEqBase + 4 * k + 2) * NS + s] = pz1;
      a.st64[(int64_t)(kEqBase + 4 * k + 3) * NS + s] = pz2;
    }
  }
}

}  // namespace af
