// af_deesser.hip -- the three-band dynamic de-esser (rust-core/src/dsp/deesser.rs) as its own pass.
//
// The de-esser is a per-sample feedback system around three dynamic peaking EQs whose RBJ
// coefficients are recomputed whenever the smoothed reduction moved by more than 0.001 dB
// (deesser.rs:536-538): ~70 f64 values of state per stream and a data-dependent coefficient update in
// the loop.  It is off by default (deesser.rs:123) and keeps its own kernel so that the default
// chain kernels stay inside their register/LDS budgets: one wavefront owns 64 streams, lane = stream,
// audio moves through a 64 x 64 LDS tile (transposed on the way in for stream-major buffers, so
// every HBM access is a coalesced 256 B row), state lives in registers for the whole launch.
//
// When the de-esser leads the chain (the default order, routing.rs eq_before_deesser = false)
// this pass also performs the input scrub / clamp / DC block / 80 Hz high-pass and the block input
// statistics, because those precede it in the reference; the chain kernel then runs without them.
#include <hip/hip_runtime.h>

#include "af_deesser_math.h"
#include "af_dsp.h"

namespace af {

using namespace deess;

struct DeEsserArgs {
  const ChainParams *params;
  double *st64;
  float *st32;
  const float *in;
  float *out;
  BlockStats *rows;  // side rows: deesser_gr_db, and input_* / output_square_sum when asked for
  int64_t n_samples, stream_stride;
  int32_t n_streams, layout;
  int32_t front_end;        // 1: input scrub/clamp/stats + DC block/high-pass happen here
  int32_t write_out_power;  // 1: rows[].output_square_sum = block power of the de-esser output
};

__global__ __launch_bounds__(kLanes) void deesser_lane_kernel(DeEsserArgs a) {
  __shared__ float x[kTile][kLanes + 1];
  const ChainParams &P = *a.params;
  const DeEsserParams &D = P.deesser;
  const int lane = threadIdx.x;
  const int s0 = blockIdx.x * kLanes;
  const int s = s0 + lane;
  const bool valid = s < a.n_streams;
  const int64_t NS = a.n_streams;
  const int sc = valid ? s : a.n_streams - 1;
  const uint32_t flags = P.flags;

  double broadband_env = a.st64[(int64_t)kDeBroadbandEnv * NS + sc];
  BandState B[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double *p = &a.st64[(int64_t)(kDeBand0 + i * kDeBandStride) * NS + sc];
    B[i].env = p[0];
    B[i].confidence = p[NS];
    B[i].baseline = p[2 * NS];
    B[i].reduction = p[3 * NS];
    B[i].gain_db = p[4 * NS];
    B[i].cancelled = p[5 * NS];
    B[i].dyn = BiquadCoef{p[6 * NS], p[7 * NS], p[8 * NS], p[9 * NS], p[10 * NS]};
    B[i].hp = Bq{p[11 * NS], p[12 * NS], p[13 * NS], p[14 * NS]};
    B[i].lp = Bq{p[15 * NS], p[16 * NS], p[17 * NS], p[18 * NS]};
    B[i].eq = Bq{p[19 * NS], p[20 * NS], p[21 * NS], p[22 * NS]};
  }
  // The front-end memories belong to whoever runs the DC block: with the suppressor on that is its pre-pass, which
  // may already be windows ahead of this launch -- touch them only when this pass filters (as the chain kernels do).
  const bool owns_front_state = a.front_end && (flags & kFlagDcBlock);
  float dc_x1 = 0.0f, dc_y1 = 0.0f;
  double pre_z1 = 0.0, pre_z2 = 0.0;
  if (owns_front_state) {
    dc_x1 = a.st32[(int64_t)kDcX1 * NS + sc];
    dc_y1 = a.st32[(int64_t)kDcY1 * NS + sc];
    pre_z1 = a.st64[(int64_t)kPreZ1 * NS + sc];
    pre_z2 = a.st64[(int64_t)kPreZ2 * NS + sc];
  }

  // deesser.rs:445-451: the `auto` curve
  const double amount = dclamp(D.auto_amount, 0.0, 1.0);
  const double trigger_offset_db = lerp(8.0, 0.8, amount);
  const double slope = lerp(0.08, 1.9, amount);
  const double auto_cap = lerp(0.8, 14.0, amount);
  const double confidence_floor = lerp(0.28, 0.06, amount);
  const double cap_db = fmin(auto_cap, D.max_reduction_db * 0.75);
  const double det_a = D.detector_attack_coeff, det_r = D.detector_release_coeff;

  const int cb = P.control_block;
  int64_t done = 0;
  int64_t block_index = 0;
  double current_reduction = 0.0;
  for (int64_t blk0 = 0; blk0 < a.n_samples; blk0 += cb, ++block_index) {
    const int blk_len = (int)((a.n_samples - blk0) < cb ? (a.n_samples - blk0) : cb);
    float in_peak = 0.0f;
    double in_sq = 0.0, out_sq = 0.0;
    for (int t0 = 0; t0 < blk_len; t0 += kTile) {
      const int len = (blk_len - t0) < kTile ? (blk_len - t0) : kTile;
      const int64_t abs0 = blk0 + t0;
      if (a.layout == 0) {
        for (int r = 0; r < kLanes; ++r) {
          const int sr = s0 + r;
          float v = 0.0f;
          if (sr < a.n_streams && lane < len) v = a.in[(int64_t)sr * a.stream_stride + abs0 + lane];
          x[lane][r] = v;
        }
      } else {
        for (int t = 0; t < len; ++t) x[t][lane] = valid ? a.in[(abs0 + t) * a.stream_stride + s] : 0.0f;
      }
      __syncthreads();

      for (int t = 0; t < len; ++t) {
        float input = x[t][lane];
        if (a.front_end) {
          // python_api.rs:515-523, routing.rs:802-843
          if ((flags & (kFlagInputScrub | kFlagInputClamp)) && !finite_f32(input)) input = 0.0f;
          if (flags & kFlagInputClamp) input = fclamp(input, -1.0f, 1.0f);
          in_sq += (double)input * (double)input;
          in_peak = fmaxf(in_peak, fabsf(input));
          if (flags & kFlagDcBlock) {
            const float o = input - dc_x1 + 0.995f * dc_y1;
            dc_x1 = input;
            dc_y1 = o;
            input = o;
            if (flags & kFlagPreHighpass) input = (float)direct(P.pre_hp, (double)o, pre_z1, pre_z2);
          }
        }
        const int64_t n = done + t;

        // ---- detector, deesser.rs:405-443
        broadband_env = smooth_value(broadband_env, (double)fabsf(input), det_a, det_r);
        double level_db[3];
        double total_env = 0.0, max_env = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const DeEsserBandParams &bp = D.bands[i];
          const int rem_hp = bp.detector_hp.xf_remaining > n ? (int)(bp.detector_hp.xf_remaining - n) : 0;
          const int rem_lp = bp.detector_lp.xf_remaining > n ? (int)(bp.detector_lp.xf_remaining - n) : 0;
          const float sc_hp = section_sample(bp.detector_hp, rem_hp, input, B[i].hp);
          const float side = section_sample(bp.detector_lp, rem_lp, sc_hp, B[i].lp);
          B[i].env = smooth_value(B[i].env, (double)fabsf(side), det_a, det_r);
          total_env += B[i].env;
          max_env = fmax(max_env, B[i].env);
          level_db[i] = lin2db(B[i].env, 1e-10);
        }
        const double voice_level = fmax(broadband_env - total_env * kVoiceRefDiscount, 1e-8);
        const double voice_db = lin2db(voice_level, 1e-10);
        const double narrowness = total_env > 1e-10 ? max_env / total_env : 0.0;

        // ---- per-band reduction targets, deesser.rs:453-517
        double target[3];
        double target_sum = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double side_db = level_db[i];
          const double ratio_db = fmax(side_db - voice_db, 0.0);
          const double dominance = max_env > 1e-10 ? sqrt(B[i].env / max_env) : 0.0;
          const double ct = confidence_target(side_db, voice_db, narrowness) * dominance;
          B[i].confidence = smooth_value(B[i].confidence, dclamp(ct, 0.0, 1.0), det_a, det_r);
          double tr = 0.0;
          if (D.auto_enabled) {
            const bool voice_active = voice_db > -55.0 || side_db > -55.0;
            if (voice_active) {
              const double bt = dclamp(ratio_db * 0.45, 0.0, 24.0);
              const double bc = bt < B[i].baseline ? D.baseline_fall : D.baseline_rise;
              B[i].baseline = bc * B[i].baseline + (1.0 - bc) * bt;
            } else {
              B[i].baseline *= D.baseline_inactive;
            }
            const double cg = normalize_range(B[i].confidence, dclamp(confidence_floor, 0.0, 0.95), 1.0);
            const double over = fmax(ratio_db - B[i].baseline - trigger_offset_db, 0.0);
            tr = dclamp(over * slope * cg, 0.0, cap_db);
          } else if (side_db > D.threshold_db) {
            const double ratio_threshold = dclamp((D.threshold_db + 60.0) * 0.10, 0.0, 6.0);
            const double level_over = side_db - D.threshold_db;
            const double ratio_over = ratio_db - ratio_threshold;
            if (ratio_over > 0.0) {
              const double over = fmin(level_over, ratio_over);
              const double cg = normalize_range(B[i].confidence, 0.22, 1.0);
              tr = dclamp((1.0 - (1.0 / D.ratio)) * over * cg, 0.0, D.max_reduction_db * 0.75);
            }
          }
          target[i] = tr;
          target_sum += tr;
        }
        if (target_sum > D.max_reduction_db && target_sum > 0.0) {
          const double scale = D.max_reduction_db / target_sum;
#pragma unroll
          for (int i = 0; i < 3; ++i) target[i] *= scale;
        }

        // ---- dynamic EQs, deesser.rs:526-546
        float processed = input;
        double total_reduction = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const DeEsserBandParams &bp = D.bands[i];
          B[i].reduction = smooth_value(B[i].reduction, target[i], D.attack_coeff, D.release_coeff);
          total_reduction += B[i].reduction;
          const double gain = -B[i].reduction;
          if (fabs(B[i].gain_db - gain) > 0.001) {  // set_gain_db_immediate -> set_coefficients_immediate
            B[i].gain_db = gain;
            B[i].dyn = peaking(bp.dyn_cos_omega, bp.dyn_alpha, gain);
            B[i].cancelled = 1.0;
            B[i].eq.pz1 = 0.0;
            B[i].eq.pz2 = 0.0;
          }
          const int rem = bp.dynamic_eq.xf_remaining > n ? (int)(bp.dynamic_eq.xf_remaining - n) : 0;
          if (rem > 0 && B[i].cancelled == 0.0) {
            processed = section_sample(bp.dynamic_eq, rem, processed, B[i].eq);
            if (rem == 1) B[i].dyn = bp.dynamic_eq.pending;
          } else {
            processed = (float)direct(B[i].dyn, (double)processed, B[i].eq.z1, B[i].eq.z2);
          }
        }
        current_reduction = fmin(total_reduction, D.max_reduction_db);
        out_sq += (double)processed * (double)processed;
        x[t][lane] = processed;
      }
      __syncthreads();
      if (a.layout == 0) {
        for (int r = 0; r < kLanes; ++r) {
          const int sr = s0 + r;
          if (sr < a.n_streams && lane < len) a.out[(int64_t)sr * a.stream_stride + abs0 + lane] = x[lane][r];
        }
      } else if (valid) {
        for (int t = 0; t < len; ++t) a.out[(abs0 + t) * a.stream_stride + s] = x[t][lane];
      }
      __syncthreads();
      done += len;
    }
    if (valid && a.rows) {
      BlockStats &row = a.rows[block_index * NS + s];
      row.deesser_gr_db = (float)current_reduction;
      if (a.front_end) {
        row.input_square_sum = in_sq;
        row.input_sample_peak = in_peak;
      }
      if (a.write_out_power) row.output_square_sum = out_sq;
    }
  }

  if (valid) {
    a.st64[(int64_t)kDeBroadbandEnv * NS + s] = broadband_env;
    a.st64[(int64_t)kDeCurrentReduction * NS + s] = current_reduction;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double *p = &a.st64[(int64_t)(kDeBand0 + i * kDeBandStride) * NS + s];
      const double v[23] = {B[i].env,    B[i].confidence, B[i].baseline, B[i].reduction, B[i].gain_db, B[i].cancelled,
                            B[i].dyn.b0, B[i].dyn.b1,     B[i].dyn.b2,   B[i].dyn.a1,    B[i].dyn.a2,  B[i].hp.z1,
                            B[i].hp.z2,  B[i].hp.pz1,     B[i].hp.pz2,   B[i].lp.z1,     B[i].lp.z2,   B[i].lp.pz1,
                            B[i].lp.pz2, B[i].eq.z1,      B[i].eq.z2,    B[i].eq.pz1,    B[i].eq.pz2};
#pragma unroll
      for (int k = 0; k < 23; ++k) p[k * NS] = v[k];
    }
    if (owns_front_state) {
      a.st32[(int64_t)kDcX1 * NS + s] = dc_x1;
      a.st32[(int64_t)kDcY1 * NS + s] = dc_y1;
      a.st64[(int64_t)kPreZ1 * NS + s] = pre_z1;
      a.st64[(int64_t)kPreZ2 * NS + s] = pre_z2;
    }
  }
}

hipError_t launch_deesser(const ChainParams *d_params, double *st64, float *st32, const float *in, float *out,
                          BlockStats *rows, int64_t n_samples, int64_t stream_stride, int32_t n_streams,
                          int32_t layout, bool front_end, bool write_out_power, hipStream_t stream) {
  DeEsserArgs a{d_params, st64, st32, in, out, rows, n_samples, stream_stride, n_streams, layout,
                front_end ? 1 : 0, write_out_power ? 1 : 0};
  const int groups = (n_streams + kLanes - 1) / kLanes;
  hipLaunchKernelGGL(deesser_lane_kernel, dim3(groups), dim3(kLanes), 0, stream, a);
  return hipGetLastError();
}

// Merge the rows the side passes produced into the chain's block-statistics rows.
__global__ void merge_side_stats_kernel(BlockStats *rows, const BlockStats *input_rows, const BlockStats *deesser_rows,
                                        int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    if (input_rows) {
      rows[i].input_square_sum = input_rows[i].input_square_sum;
      rows[i].input_sample_peak = input_rows[i].input_sample_peak;
    }
    if (deesser_rows) rows[i].deesser_gr_db = deesser_rows[i].deesser_gr_db;
  }
}
hipError_t launch_merge_side_stats(BlockStats *rows, const BlockStats *input_rows, const BlockStats *deesser_rows,
                                   int64_t n, hipStream_t stream) {
  hipLaunchKernelGGL(merge_side_stats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rows,
                     input_rows, deesser_rows, n);
  return hipGetLastError();
}

}  // namespace af
