// af_gate.hip -- the noise gate (rust-core/src/dsp/gate.rs) as the reference's offline operator uses it:
// `simulate_gate_suppressor_order` (python_api.rs:288-376) builds `NoiseGate::new(..)` + `set_gate_mode(VadAssisted)`
// without attaching a VadAutoGate, so `process_block_inplace` takes the per-sample downward-expander path
// (gate.rs:626-637): 8 ms RMS detector in dB, threshold / 4 dB hysteresis / 50 ms hold, 4:1 expansion capped at
// 36 dB (24 dB while the chatter auto-relax is active), one-pole attack/release on the linear gain, and the
// chatter detector (>= 4 open/close flips inside 500 ms, 1 s cool-down).  One recurrence per stream:
// lane per stream over 64 x 64 LDS tiles, state in registers.
#include <hip/hip_runtime.h>

#include "af_dsp.h"

namespace af {

struct GateArgs {
  const float *in;
  float *out;
  float *gain_trace;      // [block][stream] current_gain at the end of every `block` samples, or null
  uint64_t *chatter;      // [stream] chatter events of this call (added to the state's count), or null
  double *state;          // [12][stream] persistent state, or null for a one-shot from the initial state
  double threshold_db, attack_coeff, release_coeff, rms_coeff;
  int64_t n_samples, stride;
  int32_t n_streams, block, vad_mode;
  int32_t hold_samples, window_samples, cooldown_samples, relax_samples;
};

__global__ __launch_bounds__(kLanes) void gate_lane_kernel(GateArgs a) {
  __shared__ float x[kTile][kLanes + 1];
  const int lane = threadIdx.x;
  const int s0 = blockIdx.x * kLanes;
  const int s = s0 + lane;
  const bool valid = s < a.n_streams;
  double rms = 0.0, gain = 0.0;
  int hold = 0, window = 0, cooldown = 0, relax = 0, transitions = 0;
  bool open = false, eff_open = false, has_eff = false;
  uint64_t events = 0;
  const int64_t NS = a.n_streams;
  if (a.state && valid) {
    rms = a.state[0 * NS + s]; gain = a.state[1 * NS + s];
    hold = (int)a.state[2 * NS + s]; window = (int)a.state[3 * NS + s]; cooldown = (int)a.state[4 * NS + s];
    relax = (int)a.state[5 * NS + s]; transitions = (int)a.state[6 * NS + s];
    open = a.state[7 * NS + s] != 0.0; eff_open = a.state[8 * NS + s] != 0.0; has_eff = a.state[9 * NS + s] != 0.0;
  }
  int in_block = 0;
  int64_t block = 0;
  for (int64_t t0 = 0; t0 < a.n_samples; t0 += kTile) {
    const int len = (int)((a.n_samples - t0) < kTile ? (a.n_samples - t0) : kTile);
    for (int r = 0; r < kLanes; ++r) {
      const int sr = s0 + r;
      float v = 0.0f;
      if (sr < a.n_streams && lane < len) v = a.in[(int64_t)sr * a.stride + t0 + lane];
      x[lane][r] = v;
    }
    __syncthreads();
    for (int t = 0; t < len; ++t) {
      const double xin = (double)x[t][lane];
      // update_detector, gate.rs:265-285
      rms = a.rms_coeff * rms + (1.0 - a.rms_coeff) * xin * xin;
      const double level = lin2db(sqrt(rms), 1e-10);
      if (level >= a.threshold_db) {
        open = true;
        hold = a.hold_samples;
      } else if (hold > 0) {
        hold -= 1;
        open = true;
      } else if (level <= a.threshold_db - 4.0) {
        open = false;
      }
      // detector_gain_reduction_db, gate.rs:288-306
      const double range = relax > 0 ? 24.0 : 36.0;
      const double gr = open ? 0.0 : dclamp((a.threshold_db - level) * (1.0 - 1.0 / 4.0), 0.0, range);
      // track_gate_transition, gate.rs:578-611
      if (!has_eff) {
        eff_open = open;
        has_eff = true;
      } else if (open != eff_open) {
        eff_open = open;
        if (window == 0) {
          window = a.window_samples;
          transitions = 1;
        } else {
          transitions += 1;
        }
        if (transitions >= 4 && cooldown == 0) {
          events += 1;
          cooldown = a.cooldown_samples;
          if (a.vad_mode) relax = a.relax_samples;
          window = 0;
          transitions = 0;
        }
      }
      if (relax > 0) relax -= 1;  // advance_chatter_timers, gate.rs:562-575
      if (window > 0) {
        window -= 1;
        if (window == 0) transitions = 0;
      }
      if (cooldown > 0) cooldown -= 1;
      // apply_gain, gate.rs:613-623
      const double target = db2lin(-gr);
      const double coeff = target > gain ? a.attack_coeff : a.release_coeff;
      gain = coeff * gain + (1.0 - coeff) * target;
      x[t][lane] = (float)(xin * gain);
      if (++in_block == a.block) {
        if (valid && a.gain_trace) a.gain_trace[block * NS + s] = (float)gain;
        in_block = 0;
        ++block;
      }
    }
    __syncthreads();
    for (int r = 0; r < kLanes; ++r) {
      const int sr = s0 + r;
      if (sr < a.n_streams && lane < len) a.out[(int64_t)sr * a.stride + t0 + lane] = x[lane][r];
    }
    __syncthreads();
  }
  if (valid) {
    if (in_block > 0 && a.gain_trace) a.gain_trace[block * NS + s] = (float)gain;
    if (a.chatter) a.chatter[s] = events;
    if (a.state) {
      a.state[0 * NS + s] = rms; a.state[1 * NS + s] = gain;
      a.state[2 * NS + s] = hold; a.state[3 * NS + s] = window; a.state[4 * NS + s] = cooldown;
      a.state[5 * NS + s] = relax; a.state[6 * NS + s] = transitions;
      a.state[7 * NS + s] = open ? 1.0 : 0.0; a.state[8 * NS + s] = eff_open ? 1.0 : 0.0; a.state[9 * NS + s] = has_eff ? 1.0 : 0.0;
    }
  }
}

hipError_t launch_gate(const GateArgs &a, hipStream_t stream) {
  hipLaunchKernelGGL(gate_lane_kernel, dim3((a.n_streams + kLanes - 1) / kLanes), dim3(kLanes), 0, stream, a);
  return hipGetLastError();
}

}  // namespace af
