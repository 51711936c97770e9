// af_roles.hip -- Kernel 5: the dynamics chain as WAVE ROLES inside one workgroup per 64-stream group ("role pipeline").
//
// What the two earlier designs showed:
//   * token ring (af_ring_kernel.hip): every wave runs the whole chain for its 4-sample chunk and takes a token for each
//     recurrence.  A recurrence's state makes an LDS round trip per chunk and its dependent chain shares a SIMD's issue slots
//     with three other waves, so a unit is HELD 2 000-3 000 cycles per chunk where its arithmetic needs ~500: the launch is
//     paced by its longest-held unit (~875 cycles per sample step), whatever else is taken out of the kernel.
//   * stage pipeline (af_stages.hip): every recurrence is a wave of its own with its state in REGISTERS and nothing else in
//     its loop -- 100-170 cycles per sample step -- but every hand-over goes through rings in HBM (0.3 KB per sample step
//     and stream), which is what a large batch then waits for.
// Here a recurrence is still a dedicated wave with register state, the feed-forward math between two recurrences is done
// by other waves of the SAME workgroup, and every hand-over is a tile of kRT sample steps x 64 streams in LDS: one workgroup
// barrier per tile, role k works on tile (iteration - depth k).  No tokens, no polling, no HBM between stages.
//
// The chain is two such kernels, cut where the compressor hands its output to the limiter (4 bytes per sample through HBM,
// in place in the caller's buffer): `chain_comp_roles_kernel` (side-chain filters and envelopes | detector levels | peak
// envelope | gain-reduction target | gain-reduction smoothing + makeup | gain) and `chain_lim_roles_kernel` (lookahead limiter |
// input-side 4x true peak | true-peak gain, output, block statistics | output-side 4x true peak).  Both read and write the
// state planes of the token-ring kernel (af_device.h), so the kernels can be mixed mid-stream; a configuration these
// kernels do not build stays on the token ring (the host decides, af_api.cpp).  The EQ and the block input statistics are
// the systolic EQ kernel's (af_eq_systolic.hip).
//
// Arithmetic: the expressions are the token-ring kernel's, operation for operation (no contraction), in the same order per
// stream -- audio, block rows and state agree bit for bit (tests/test_gpu_roles.py).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "af_dsp.h"

namespace af {
namespace {

// Workgroup barrier for LDS hand-over only: waits for this wave's LDS traffic, not for its global loads and stores (a
// __syncthreads() also drains those -- the audio fetched one iteration ahead would be waited for at every tile).
__device__ __forceinline__ void lds_barrier() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kRT = 8;  // sample steps per tile = per workgroup barrier (a control block must be a whole number of tiles)

// ---------------------------------------------------------------------------------------------------------- compressor
// waves: 0 = A (side-chain high-pass, band / rms envelopes), 1 = C (peak envelope), 2 = E (gain-reduction smoothing, makeup),
//        3..10 = feed-forward: wave 3 + k takes step k of every tile through F1 (detector levels), F2 (gain-reduction target)
//        and the gain; waves 3 and 7 also load and store the audio (four steps per lane: one 16-byte access)
constexpr int kCompWaves = 11;
constexpr int kCompDepth = 6;   // the gain stage works six tiles behind the load
constexpr int kXRing = 8;       // input tiles kept in LDS: written at depth 0, read by A at 1, turned into the output at 6, stored at 7

struct CompLds {
  float X[kXRing][kRT][kLanes];
  double D[2][kRT][kLanes], LOW[2][kRT][kLanes], VOI[2][kRT][kLanes], PRES[2][kRT][kLanes], RMS[2][kRT][kLanes];  // A -> F1
  double IPK[2][kRT][kLanes];                              // F1 -> C
  double RMSDB[4][kRT][kLanes], WDB[4][kRT][kLanes];       // F1 -> F2 (two tiles later)
  double PEAK[2][kRT][kLanes];                             // C -> F2
  double TARGET[2][kRT][kLanes];                           // F2 -> E
  double GR[2][kRT][kLanes];                               // E -> gain
  double MK[2][kLanes];                                    // E -> gain: the linear makeup gain in force during the tile
  double PLOS[kLanes];                                     // diagnostic state: the last step's plosive ratio
};

template <bool kSc, bool kAdaptive>
__global__ __launch_bounds__(64 * kCompWaves) void chain_comp_roles_kernel(LaunchArgs a, const ChainParams *__restrict__ params) {
  extern __shared__ __attribute__((aligned(16))) unsigned char roles_lds[];
  CompLds &L = *reinterpret_cast<CompLds *>(roles_lds);
  const ChainParams &P = params[a.group_preset ? a.group_preset[blockIdx.x] : 0];  // the preset of this 64-stream group
  const int tid = threadIdx.x, lane = tid & (kLanes - 1), wave = tid >> 6;
  const int s = blockIdx.x * kLanes + lane;
  const bool valid = s < a.n_streams;
  const int sc = valid ? s : a.n_streams - 1;
  const int64_t NS = a.n_streams;
  const int64_t n = a.n_samples;
  const int64_t ntiles = (n + kRT - 1) / kRT;
  const int cb = P.control_block;
  const bool vec_ok = a.layout == 0 && (a.stream_stride % 4) == 0 &&
                      ((reinterpret_cast<uintptr_t>(a.in) | reinterpret_cast<uintptr_t>(a.out)) & 15) == 0;
  auto tile_len = [&](int64_t ti) { return (int)((n - ti * kRT) < kRT ? (n - ti * kRT) : kRT); };

  // ---- per-role state (registers) and the parameters its loop reads (copied out of the block once)
  // A
  double prev_in = 0.0, prev_out = 0.0, low_env = 0.0, voiced_env = 0.0, presence_env = 0.0, rms_env = 0.0;
  // C
  double pe = 0.0;
  // E
  double gr = 0.0, fast = 0.0, slow = 0.0, cur_ms = 0.0, tgt_ms = 0.0, rel_coeff = 0.0, sm = 0.0, makeup_lin = 1.0;
  const CompressorParams cp = P.comp;  // by value: read through the pointer a field would be re-loaded after every LDS store
  if (wave < 3) __builtin_amdgcn_s_setprio(3);  // a recurrence's dependent chain goes first: the feed-forward waves fill the gaps
  if (wave == 0) {
    rms_env = a.st64[(int64_t)kCompRmsEnvSq * NS + sc];
    if (kSc) {
      prev_in = a.st64[(int64_t)kCompScPrevIn * NS + sc];
      prev_out = a.st64[(int64_t)kCompScPrevOut * NS + sc];
      low_env = a.st64[(int64_t)kCompLowEnv * NS + sc];
      voiced_env = a.st64[(int64_t)kCompVoicedEnv * NS + sc];
      presence_env = a.st64[(int64_t)kCompPresenceEnv * NS + sc];
    }
  } else if (wave == 1) {
    pe = a.st64[(int64_t)kCompPeakEnvDb * NS + sc];
  } else if (wave == 2) {
    gr = a.st64[(int64_t)kCompGr * NS + sc];
    fast = a.st64[(int64_t)kCompFastEnv * NS + sc];
    slow = a.st64[(int64_t)kCompSlowEnv * NS + sc];
    cur_ms = a.st64[(int64_t)kCompCurReleaseMs * NS + sc];
    tgt_ms = a.st64[(int64_t)kCompTargetReleaseMs * NS + sc];
    rel_coeff = a.st64[(int64_t)kCompReleaseCoeff * NS + sc];
    sm = a.st64[(int64_t)kCompSmoothedMakeup * NS + sc];
    makeup_lin = db2lin(sm);
  }
  if (wave == 0) L.PLOS[lane] = 0.0;
  float pre[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // waves 3 and 7: the four steps of the next tile, in flight
  if (wave >= 3 && ((wave - 3) & 3) == 0 && ntiles > 0) {
    const int w4 = wave - 3;
    const int len = tile_len(0);
    if (vec_ok && w4 + 4 <= len) {
      const float4 v = valid ? *reinterpret_cast<const float4 *>(&a.in[(int64_t)s * a.stream_stride + w4]) : make_float4(0, 0, 0, 0);
      pre[0] = v.x; pre[1] = v.y; pre[2] = v.z; pre[3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (valid && w4 + j < len) pre[j] = a.layout == 0 ? a.in[(int64_t)s * a.stream_stride + w4 + j] : a.in[(int64_t)(w4 + j) * a.stream_stride + s];
    }
  }

  for (int64_t it = 0; it < ntiles + kCompDepth + 2; ++it) {
    if (wave == 0) {
      // =================== A: side-chain high-pass + band / rms envelopes (compressor.rs:700-733), tile it - 1
      const int64_t ti = it - 1;
      if (ti >= 0 && ti < ntiles) {
        const int len = tile_len(ti);
        const float(&X)[kRT][kLanes] = L.X[ti & (kXRing - 1)];
        const int b = (int)(ti & 1);
        const double kk = cp.band_env_coeff;
#pragma unroll
        for (int t = 0; t < kRT; ++t)
          if (t < len) {
            const double xin = (double)X[t][lane];
            if (kSc) {
              const double dd = cp.sidechain_highpass_coeff * (prev_out + xin - prev_in);
              prev_in = xin;
              prev_out = dd;
              const double low = xin - dd;
              const double presence = 0.65 * dd + 0.35 * (dd - low);
              low_env = kk * low_env + (1.0 - kk) * low * low;
              voiced_env = kk * voiced_env + (1.0 - kk) * dd * dd;
              presence_env = kk * presence_env + (1.0 - kk) * presence * presence;
              rms_env = cp.rms_coeff * rms_env + (1.0 - cp.rms_coeff) * (dd * dd);
              L.D[b][t][lane] = dd;
              L.LOW[b][t][lane] = low_env;
              L.VOI[b][t][lane] = voiced_env;
              L.PRES[b][t][lane] = presence_env;
              L.RMS[b][t][lane] = rms_env;
            } else {
              const double dd = xin;
              rms_env = cp.rms_coeff * rms_env + (1.0 - cp.rms_coeff) * (dd * dd);
              L.D[b][t][lane] = dd;
              L.RMS[b][t][lane] = rms_env;
            }
          }
      }
    } else if (wave == 1) {
      // =================== C: log-domain peak envelope (compressor.rs:735-742), tile it - 3
      const int64_t ti = it - 3;
      if (ti >= 0 && ti < ntiles) {
        const int len = tile_len(ti);
        const int b = (int)(ti & 1);
#pragma unroll
        for (int t = 0; t < kRT; ++t)
          if (t < len) {
            const double v = L.IPK[b][t][lane];
            const double pk = v > pe ? cp.attack_coeff : cp.detector_release_coeff;
            pe = pk * pe + (1.0 - pk) * v;
            L.PEAK[b][t][lane] = pe;
          }
      }
    } else if (wave == 2) {
      // =================== E: release-time meter + gain-reduction smoothing, makeup per control block
      // (compressor.rs:452-505,604-617,752-764), tile it - 5
      const int64_t ti = it - 5;
      if (ti >= 0 && ti < ntiles) {
        const int len = tile_len(ti);
        const int b = (int)(ti & 1);
        L.MK[b][lane] = makeup_lin;  // the gain in force during this tile (a control block ends on a tile boundary)
#pragma unroll
        for (int t = 0; t < kRT; ++t)
          if (t < len) {
            if (kAdaptive) {
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
            const double tg = L.TARGET[b][t][lane];
            if (!kAdaptive) {
              const double k2 = tg > gr ? cp.attack_coeff : rel_coeff;
              gr = k2 * gr + (1.0 - k2) * tg;
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
            L.GR[b][t][lane] = gr;
          }
        const int64_t t_end = ti * kRT + len;
        if (t_end % cb == 0 || t_end == n) {  // (wave-uniform) a control block ends with this tile
          const int64_t blk = (t_end - 1) / cb;
          const int blk_len = (int)(t_end - blk * cb);
          // update_auto_makeup_gain with auto-makeup off (compressor.rs:604-617)
          const double makeup_coeff = pow(cp.makeup_smoothing_coeff, (double)(blk_len < 1 ? 1 : blk_len));
          const double tgt = cp.makeup_gain_db;
          if (fabs(tgt - sm) > 0.1) {
            sm = makeup_coeff * sm + (1.0 - makeup_coeff) * tgt;
          } else {
            sm = tgt;
          }
          makeup_lin = db2lin(sm);
          if (valid && a.stats) {
            BlockStats &row = a.stats[blk * NS + s];
            row.compressor_gr_db = (float)gr;
            row.makeup_gain_db = (float)sm;
          }
        }
      }
    } else {
      // =================== feed-forward waves 3..10: wave k takes step t = k of three tiles at once -- F1 of tile it - 2,
      // F2 of tile it - 4, the gain of tile it - 6 -- so every SIMD carries the same share of the f64 work (what bounds this
      // kernel: ~530 f64 instructions per step and stream group); waves 3 and 7 also move the audio, four steps per lane as
      // one 16-byte access, the load one iteration ahead in registers
      const int t = wave - 3;
      {
        // ---- F1: detector weight, instantaneous peak and RMS levels in dB (update_sidechain_band_metrics, compressor.rs:438-449)
        const int64_t ti = it - 2;
        if (ti >= 0 && ti < ntiles && t < tile_len(ti)) {
          const int b = (int)(ti & 1), b4 = (int)(ti & 3);
          double weight_db = kDetectorUnitWeight, plosive_last = 0.0;  // (WDB / RMSDB: dB in the literal build, linear otherwise)
          if (kSc) {
            const double low_rms = sqrt(L.LOW[b][t][lane]);
            const double voiced_rms = fmax(sqrt(L.VOI[b][t][lane]), 1e-8);
            const double presence_rms = sqrt(L.PRES[b][t][lane]);
            const double plosive = dclamp(low_rms / voiced_rms, 0.0, 32.0);
            plosive_last = plosive;
            const double plosive_amount = dclamp(div_known(plosive - 1.25, 3.75, 1.0 / 3.75), 0.0, 1.0);
            const double plosive_penalty = 1.0 - plosive_amount * (1.0 - 0.35);
            const double presence_ratio = dclamp(presence_rms / voiced_rms, 0.0, 4.0);
            const double presence_weight = 1.0 + 0.18 * dclamp(presence_ratio - 0.75, 0.0, 1.0);
            weight_db = detector_weight(dclamp(plosive_penalty * presence_weight, 0.35, 1.15));
          }
          L.WDB[b4][t][lane] = weight_db;
          L.IPK[b][t][lane] = lin2db(fabs(L.D[b][t][lane]), 1e-10);
          L.RMSDB[b4][t][lane] = detector_rms_level(L.RMS[b][t][lane]);
          if (ti * kRT + t == n - 1) L.PLOS[lane] = plosive_last;
        }
      }
      {
        // ---- F2: blended detector level -> static gain-reduction target (compressor.rs:744-750,657-678)
        const int64_t ti = it - 4;
        if (ti >= 0 && ti < ntiles && t < tile_len(ti)) {
          const int b = (int)(ti & 1), b4 = (int)(ti & 3);
          L.TARGET[b][t][lane] = comp_gain_reduction(cp, detector_db(L.PEAK[b][t][lane], L.RMSDB[b4][t][lane], L.WDB[b4][t][lane]));
        }
      }
      {
        // ---- gain (compressor.rs:771-773): the result goes back into the tile's X slot (its last reader), stored below
        const int64_t ti = it - kCompDepth;
        if (ti >= 0 && ti < ntiles && t < tile_len(ti)) {
          const int b = (int)(ti & 1);
          float(&X)[kRT][kLanes] = L.X[ti & (kXRing - 1)];
          X[t][lane] = (float)((double)X[t][lane] * (db2lin(-L.GR[b][t][lane]) * L.MK[b][lane]));
        }
      }
      if ((t & 3) == 0) {
        // ---- audio in: tile `it` leaves the registers it was fetched into during the previous iteration, tile it + 1 is fetched
        const int w4 = t;  // first of this wave's four steps
        if (it < ntiles) {
          float(&X)[kRT][kLanes] = L.X[it & (kXRing - 1)];
#pragma unroll
          for (int j = 0; j < 4; ++j) X[w4 + j][lane] = pre[j];
        }
        auto fetch = [&](int64_t ti) {
          if (ti >= ntiles) return;
          const int len = tile_len(ti);
          const int64_t t0 = ti * kRT + w4;
          if (vec_ok && w4 + 4 <= len) {
            const float4 v = valid ? *reinterpret_cast<const float4 *>(&a.in[(int64_t)s * a.stream_stride + t0]) : make_float4(0, 0, 0, 0);
            pre[0] = v.x; pre[1] = v.y; pre[2] = v.z; pre[3] = v.w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              pre[j] = 0.0f;
              if (valid && w4 + j < len) pre[j] = a.layout == 0 ? a.in[(int64_t)s * a.stream_stride + t0 + j] : a.in[(t0 + j) * a.stream_stride + s];
            }
          }
        };
        fetch(it + 1);
        // ---- audio out: tile it - 7, whose gains the eight waves applied during the previous iteration
        const int64_t to = it - (kCompDepth + 1);
        if (to >= 0 && to < ntiles) {
          const int len = tile_len(to);
          const float(&X)[kRT][kLanes] = L.X[to & (kXRing - 1)];
          const int64_t t0 = to * kRT + w4;
          if (vec_ok && w4 + 4 <= len) {
            if (valid)
              *reinterpret_cast<float4 *>(&a.out[(int64_t)s * a.stream_stride + t0]) =
                  make_float4(X[w4][lane], X[w4 + 1][lane], X[w4 + 2][lane], X[w4 + 3][lane]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (valid && w4 + j < len) {
                if (a.layout == 0) a.out[(int64_t)s * a.stream_stride + t0 + j] = X[w4 + j][lane];
                else a.out[(t0 + j) * a.stream_stride + s] = X[w4 + j][lane];
              }
          }
        }
      }
    }
    lds_barrier();
  }

  // ---- state back to the planes (the fields the token-ring kernel keeps for these recurrences)
  if (valid && n > 0) {
    if (wave == 0) {
      a.st64[(int64_t)kCompRmsEnvSq * NS + s] = rms_env;
      if (kSc) {
        a.st64[(int64_t)kCompScPrevIn * NS + s] = prev_in;
        a.st64[(int64_t)kCompScPrevOut * NS + s] = prev_out;
        a.st64[(int64_t)kCompLowEnv * NS + s] = low_env;
        a.st64[(int64_t)kCompVoicedEnv * NS + s] = voiced_env;
        a.st64[(int64_t)kCompPresenceEnv * NS + s] = presence_env;
      }
      a.st64[(int64_t)kCompPlosive * NS + s] = L.PLOS[lane];  // (diagnostic only, compressor.rs:441)
    } else if (wave == 1) {
      a.st64[(int64_t)kCompPeakEnvDb * NS + s] = pe;
    } else if (wave == 2) {
      a.st64[(int64_t)kCompGr * NS + s] = gr;
      a.st64[(int64_t)kCompFastEnv * NS + s] = fast;
      a.st64[(int64_t)kCompSlowEnv * NS + s] = slow;
      a.st64[(int64_t)kCompCurReleaseMs * NS + s] = cur_ms;
      a.st64[(int64_t)kCompTargetReleaseMs * NS + s] = tgt_ms;
      a.st64[(int64_t)kCompSmoothedMakeup * NS + s] = sm;
      if (kAdaptive) {
        const double tau = fmax(cur_ms, 0.001) / 1000.0;  // compressor.rs:760-761
        a.st64[(int64_t)kCompReleaseCoeff * NS + s] = exp(-1.0 / (tau * cp.sample_rate));
      }
    }
  }
}


// ------------------------------------------------------------------------------------------- limiter + true-peak limiter
// waves: 0 = VH   sliding maximum over the lookahead window (van Herk: ring + suffix maxima of the previous W-aligned block,
//                 running prefix maximum of the current one -- the token-ring kernel's own arrays, also as state), tile it - 1
//        1 = LIM  limiter gain (limiter.rs:271-284), tile it - 3
//        2 = TPO  true-peak gain (true_peak.rs:353-374), chain output, block output statistics, tile it - 6
//        3 = load of tile `it` and F4: the gain the window maximum asks for (one f64 division per step), tile it - 2
//        4 = F5a: limiter output = clamp(delayed x gain) into the true-peak input ring, tile it - 4; store of tile it - 7 and the
//            block's output true peak
//        5..8 = F5b: input-side 4x true peak (4 phases x 32 taps) and the gain it asks for, tile it - 5, two steps per wave
//        9..12 = F6: output-side 4x true peak folded into the block maximum, tile it - 7, two steps per wave
constexpr int kLimWaves = 13;
constexpr int kLimDepth = 8;    // iterations a tile needs from its load to the last thing done with it
constexpr int kHRing = 64;      // rows of the true-peak input / output rings (32 taps + 20 samples of delay + the tiles in flight)

struct LimLds {
  float XC[2][kRT][kLanes];                              // load -> VH
  float PK[2][kRT][kLanes];                              // VH -> F4: window maximum
  float DL[4][kRT][kLanes];                              // VH -> F5a: the sample leaving the lookahead delay
  double TG[2][kRT][kLanes];                             // F4 -> LIM
  double G[2][kRT][kLanes];                              // LIM -> F5a
  float XL[kHRing][kLanes];                              // F5a -> F5b, TPO: true-peak limiter input, sample n in row n & 63
  float ITP[2][kRT][kLanes], TGT[2][kRT][kLanes];        // F5b -> TPO
  float O[kHRing][kLanes];                               // TPO -> F6, store: chain output, sample n in row n & 63
  unsigned int OTP[2][kLanes];                           // F6 -> store wave: the block's output true peak (bit pattern), by block parity
  // then: float RING[W][64], SUF[W][64] (dynamic)
};

__global__ __launch_bounds__(64 * kLimWaves) void chain_lim_roles_kernel(LaunchArgs a, const ChainParams *__restrict__ params) {
  extern __shared__ __attribute__((aligned(16))) unsigned char roles_lds[];
  LimLds &L = *reinterpret_cast<LimLds *>(roles_lds);
  const ChainParams &P = params[a.group_preset ? a.group_preset[blockIdx.x] : 0];
  const int W = P.lim.lookahead_samples + 1;
  float *ring = reinterpret_cast<float *>(roles_lds + sizeof(LimLds));
  float *suf = ring + (size_t)W * kLanes;
  const int tid = threadIdx.x, lane = tid & (kLanes - 1), wave = tid >> 6;
  const int s = blockIdx.x * kLanes + lane;
  const bool valid = s < a.n_streams;
  const int sc = valid ? s : a.n_streams - 1;
  const int64_t NS = a.n_streams;
  const int64_t n = a.n_samples, n0 = a.samples_before;
  const int64_t ntiles = (n + kRT - 1) / kRT;
  const int cb = P.control_block;
  const bool vec_ok = a.layout == 0 && (a.stream_stride % 4) == 0 &&
                      ((reinterpret_cast<uintptr_t>(a.in) | reinterpret_cast<uintptr_t>(a.out)) & 15) == 0;
  auto tile_len = [&](int64_t ti) { return (int)((n - ti * kRT) < kRT ? (n - ti * kRT) : kRT); };
  auto ends_block = [&](int64_t ti, int len) { const int64_t e = ti * kRT + len; return e % cb == 0 || e == n; };
  const double ceil_lin = P.lim.ceiling_linear;
  const float tp_ceiling = P.tp.ceiling_linear;

  // ---- histories into LDS: the limiter's ring and suffix maxima, the last 32 true-peak-limiter inputs and chain outputs
  for (int r = wave; r < 2 * W; r += kLimWaves) ring[(size_t)r * kLanes + lane] = a.st32[(int64_t)(kLimRing + r) * NS + sc];
  for (int r = wave; r < kTpTaps; r += kLimWaves) {
    const int row = (int)((n0 - kTpTaps + r) & (kHRing - 1));  // state row r holds sample n0 - 32 + r
    L.XL[row][lane] = a.st32[(int64_t)(kTpInHist + r) * NS + sc];
    L.O[row][lane] = a.st32[(int64_t)(kTpOutHist + r) * NS + sc];
  }
  if (wave == 0) {
    L.OTP[0][lane] = 0u;
    L.OTP[1][lane] = 0u;
  }
  if (wave < 3) __builtin_amdgcn_s_setprio(3);  // (the recurrences first)
  // per-role state
  float prefix = 0.0f;                                   // VH
  int vh_j = 0;
  double g = 1.0, gmin = 1.0;                            // LIM
  const double rc = P.lim.release_coeff;
  float tpg = 1.0f, tp_in_peak = 0.0f, tp_gmin = 1.0f, tp_limited = 0.0f, out_peak = 0.0f, nonfinite = 0.0f;  // TPO
  double out_sq = 0.0;
  const float rel = P.tp.release_coeff;
  if (wave == 0) {
    prefix = a.st32[(int64_t)kLimPrefix * NS + sc];
    vh_j = (int)(n0 % W);
  } else if (wave == 1) {
    g = a.st64[(int64_t)kLimGain * NS + sc];
  } else if (wave == 2) {
    tpg = a.st32[(int64_t)kTpGain * NS + sc];
  }
  float pre[kRT];  // wave 3: the next tile's eight steps, in flight
#pragma unroll
  for (int j = 0; j < kRT; ++j) pre[j] = 0.0f;
  auto fetch_tile = [&](int64_t ti) {
    if (ti >= ntiles) return;
    const int len = tile_len(ti);
#pragma unroll
    for (int h = 0; h < kRT; h += 4) {
      const int64_t t0 = ti * kRT + h;
      if (vec_ok && h + 4 <= len) {
        const float4 v = valid ? *reinterpret_cast<const float4 *>(&a.in[(int64_t)s * a.stream_stride + t0]) : make_float4(0, 0, 0, 0);
        pre[h] = v.x; pre[h + 1] = v.y; pre[h + 2] = v.z; pre[h + 3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pre[h + j] = 0.0f;
          if (valid && h + j < len) pre[h + j] = a.layout == 0 ? a.in[(int64_t)s * a.stream_stride + t0 + j] : a.in[(t0 + j) * a.stream_stride + s];
        }
      }
    }
  };
  if (wave == 3) fetch_tile(0);
  __syncthreads();

  for (int64_t it = 0; it < ntiles + kLimDepth + 1; ++it) {
    if (wave == 0) {
      // =================== VH (limiter.rs:246-270, the part with memory), tile it - 1
      const int64_t ti = it - 1;
      if (ti >= 0 && ti < ntiles) {
        const int len = tile_len(ti);
        const int b = (int)(ti & 1), b4 = (int)(ti & 3);
        int j = vh_j;
#pragma unroll
        for (int t = 0; t < kRT; ++t)
          if (t < len) {
            const float xin = L.XC[b][t][lane];
            const float ax = fabsf(xin);
            const int jn = (j + 1 == W) ? 0 : j + 1;
            const float delayed = ring[jn * kLanes + lane];
            const float sfx = (j + 1 < W) ? suf[(j + 1) * kLanes + lane] : 0.0f;
            prefix = (j == 0) ? ax : fmaxf(prefix, ax);
            L.PK[b][t][lane] = fmaxf(sfx, prefix);
            L.DL[b4][t][lane] = delayed;
            ring[j * kLanes + lane] = xin;
            if (j + 1 == W) {  // suffix maxima of the block just completed, eight at a time
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
          }
        vh_j = j;
      }
    } else if (wave == 1) {
      // =================== LIM: gain smoothing (limiter.rs:271-284), tile it - 3
      const int64_t ti = it - 3;
      if (ti >= 0 && ti < ntiles) {
        const int len = tile_len(ti);
        const int b = (int)(ti & 1);
#pragma unroll
        for (int t = 0; t < kRT; ++t)
          if (t < len) {
            const double tg = L.TG[b][t][lane];
            if (tg < g) {
              g = tg;
            } else {
              g = rc * g + (1.0 - rc) * tg;
            }
            gmin = fmin(gmin, g);
            L.G[b][t][lane] = g;
          }
        if (ends_block(ti, len)) {
          const int64_t blk = (ti * kRT + len - 1) / cb;
          if (valid && a.stats) a.stats[blk * NS + s].limiter_peak_gr_db = gmin < 1.0 ? (float)(-lin2db(gmin, 1e-10)) : 0.0f;
          gmin = 1.0;
        }
      }
    } else if (wave == 2) {
      // =================== TPO: true-peak gain (true_peak.rs:341-374), chain output, block output statistics
      // (block_processor.rs:150-170), tile it - 6
      const int64_t ti = it - 6;
      if (ti >= 0 && ti < ntiles) {
        const int len = tile_len(ti);
        const int b = (int)(ti & 1);
#pragma unroll
        for (int t = 0; t < kRT; ++t)
          if (t < len) {
            const int64_t na = n0 + ti * kRT + t;
            const float delayed = L.XL[(na - kTpDelay) & (kHRing - 1)][lane];
            const float itp = L.ITP[b][t][lane], tg = L.TGT[b][t][lane];
            tp_in_peak = fmaxf(tp_in_peak, itp);
            if (tg < tpg) {
              tpg = tg;
              tp_limited = 1.0f;
            } else {
              tpg = rel * tpg + (1.0f - rel) * tg;
            }
            tp_gmin = fminf(tp_gmin, tpg);
            float o = fclamp(delayed * tpg, -tp_ceiling, tp_ceiling);
            if (!finite_f32(o)) o = 0.0f;
            float det = o;
            if (finite_f32(o)) {
              out_sq += (double)o * (double)o;
            } else {
              nonfinite = 1.0f;
              det = 0.0f;  // TruePeakDetector::process_block, true_peak.rs:212
            }
            out_peak = fmaxf(out_peak, fabsf(o));
            L.O[na & (kHRing - 1)][lane] = det;
          }
        if (ends_block(ti, len)) {
          const int64_t blk = (ti * kRT + len - 1) / cb;
          if (valid && a.stats) {
            BlockStats &row = a.stats[blk * NS + s];
            row.output_square_sum = out_sq;
            row.output_sample_peak = out_peak;
            row.non_finite_output = nonfinite != 0.0f ? 1u : 0u;
            row.tp_limiter_input_peak = tp_in_peak;
            row.tp_limiter_gr_db = tp_gmin < 1.0f ? -20.0f * log10f(fmaxf(tp_gmin, 1e-10f)) : 0.0f;
            row.tp_limited_events = tp_limited != 0.0f ? 1u : 0u;
          }
          out_sq = 0.0;
          out_peak = 0.0f;
          nonfinite = 0.0f;
          tp_in_peak = 0.0f;
          tp_gmin = 1.0f;
          tp_limited = 0.0f;
        }
      }
    } else if (wave == 3) {
      // =================== audio in: tile `it` leaves the registers it was fetched into during the previous iteration, tile
      // it + 1 is fetched; F4: the gain the window maximum asks for, tile it - 2
      if (it < ntiles) {
        float(&X)[kRT][kLanes] = L.XC[it & 1];
#pragma unroll
        for (int j = 0; j < kRT; ++j) X[j][lane] = pre[j];
      }
      fetch_tile(it + 1);
      {
        const int64_t ti = it - 2;
        if (ti >= 0 && ti < ntiles) {
          const int len = tile_len(ti);
          const int b = (int)(ti & 1);
#pragma unroll
          for (int t = 0; t < kRT; ++t)
            if (t < len) {
              const double peak = (double)L.PK[b][t][lane];
              L.TG[b][t][lane] = peak > ceil_lin ? ceil_lin / peak : 1.0;
            }
        }
      }
    } else if (wave == 4) {
      // =================== F5a: limiter output (limiter.rs:278-284, scrubbed for the true-peak limiter, true_peak.rs:342),
      // tile it - 4; store of tile it - 7 and, when it ended a control block, the block's output true peak
      {
        const int64_t ti = it - 4;
        if (ti >= 0 && ti < ntiles) {
          const int len = tile_len(ti);
          const int b = (int)(ti & 1), b4 = (int)(ti & 3);
#pragma unroll
          for (int t = 0; t < kRT; ++t)
            if (t < len) {
              const float o = (float)dclamp((double)L.DL[b4][t][lane] * L.G[b][t][lane], -ceil_lin, ceil_lin);
              L.XL[(n0 + ti * kRT + t) & (kHRing - 1)][lane] = finite_f32(o) ? o : 0.0f;
            }
        }
      }
      {
        const int64_t ti = it - 7;
        if (ti >= 0 && ti < ntiles) {
          const int len = tile_len(ti);
#pragma unroll
          for (int h = 0; h < kRT; h += 4) {
            const int64_t t0 = ti * kRT + h;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = L.O[(n0 + t0 + j) & (kHRing - 1)][lane];
            if (vec_ok && h + 4 <= len) {
              if (valid) *reinterpret_cast<float4 *>(&a.out[(int64_t)s * a.stream_stride + t0]) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (valid && h + j < len) {
                  if (a.layout == 0) a.out[(int64_t)s * a.stream_stride + t0 + j] = o[j];
                  else a.out[(t0 + j) * a.stream_stride + s] = o[j];
                }
            }
          }
        }
        // the F6 waves folded tile it - 8 into its block's maximum during the previous iteration
        const int64_t tj = it - 8;
        if (tj >= 0 && tj < ntiles) {
          const int len = tile_len(tj);
          if (ends_block(tj, len)) {
            const int64_t blk = (tj * kRT + len - 1) / cb;
            if (valid && a.stats) a.stats[blk * NS + s].output_true_peak = __uint_as_float(L.OTP[blk & 1][lane]);
            L.OTP[blk & 1][lane] = 0u;
          }
        }
      }
    } else {
      // =================== F5b (waves 5..8, tile it - 5) and F6 (waves 9..12, tile it - 7): Bandlimited4xPeak::observe
      // (true_peak.rs:173-186) over a ring, two consecutive steps per wave sharing 31 of their 32 window samples
      const bool out_side = wave >= 9;
      const int64_t ti = it - (out_side ? 7 : 5);
      if (ti >= 0 && ti < ntiles) {
        const int len = tile_len(ti);
        const int t0 = ((wave - 5) & 3) * 2;
        if (t0 < len) {
          const float(*R)[kLanes] = out_side ? L.O : L.XL;
          const int64_t na = n0 + ti * kRT + t0;  // absolute index of the first of the two steps
          float h[kTpTaps + 1];                   // h[i]: sample na - 31 + i
#pragma unroll
          for (int i = 0; i < kTpTaps + 1; ++i) h[i] = R[(na - (kTpTaps - 1) + i) & (kHRing - 1)][lane];
          float pk[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            float peak = fabsf(h[kTpTaps - 1 + u]);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              float acc = 0.0f;
#pragma unroll
              for (int k = 0; k < kTpTaps; ++k) acc = __builtin_fmaf(AF_TP_FIR[p][k], h[kTpTaps - 1 + u - k], acc);
              peak = fmaxf(peak, fabsf(acc));
            }
            pk[u] = peak;
          }
          if (!out_side) {
            const int b = (int)(ti & 1);
#pragma unroll
            for (int u = 0; u < 2; ++u)
              if (t0 + u < len) {
                float tg = 1.0f;
                if (pk[u] > tp_ceiling) tg = fclamp((tp_ceiling * 0.999f) / pk[u], 0.0f, 1.0f);
                L.ITP[b][t0 + u][lane] = pk[u];
                L.TGT[b][t0 + u][lane] = tg;
              }
          } else {
            float m = pk[0];
            if (t0 + 1 < len) m = fmaxf(m, pk[1]);
            const int64_t blk = (ti * kRT + t0) / cb;  // (both steps lie in one block: a block is a whole number of tiles)
            atomicMax(&L.OTP[blk & 1][lane], __float_as_uint(m));
          }
        }
      }
    }
    lds_barrier();
  }

  // ---- state back to the planes
  if (n > 0) {
    const int64_t n_end = n0 + n;
    if (valid) {
      if (wave == 0) a.st32[(int64_t)kLimPrefix * NS + s] = prefix;
      if (wave == 1) a.st64[(int64_t)kLimGain * NS + s] = g;
      if (wave == 2) a.st32[(int64_t)kTpGain * NS + s] = tpg;
      for (int r = wave; r < 2 * W; r += kLimWaves) a.st32[(int64_t)(kLimRing + r) * NS + s] = ring[(size_t)r * kLanes + lane];
      for (int r = wave; r < kTpTaps; r += kLimWaves) {
        const int row = (int)((n_end - kTpTaps + r) & (kHRing - 1));
        a.st32[(int64_t)(kTpInHist + r) * NS + s] = L.XL[row][lane];
        a.st32[(int64_t)(kTpOutHist + r) * NS + s] = L.O[row][lane];
      }
    }
  }
}

}  // namespace

bool comp_roles_serves(const ChainParams &p) {
  return (p.flags & kFlagCompressor) && !p.comp.auto_makeup_enabled && p.control_block % kRT == 0 && !(p.flags & kFlagDeesser);
}
size_t lim_roles_lds_bytes(int lookahead_samples) { return sizeof(LimLds) + (size_t)2 * (lookahead_samples + 1) * kLanes * sizeof(float); }
bool lim_roles_serves(const ChainParams &p) {
  return (p.flags & kFlagLimiter) && p.control_block % kRT == 0 && !(p.flags & kFlagDeesser) &&
         lim_roles_lds_bytes(p.lim.lookahead_samples) <= 160 * 1024;
}

// lookahead limiter -> true-peak limiter -> output statistics and detector over `args.n_samples` steps, in -> out (may be the
// same buffer); needs lim_roles_serves().  `max_lookahead`: the largest lookahead among the presets of the launch.
hipError_t launch_chain_lim_roles(const LaunchArgs &args, int max_lookahead, hipStream_t stream) {
  const int groups = (args.n_streams + kLanes - 1) / kLanes;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(chain_lim_roles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) return err;
    attr_set = true;
  }
  hipLaunchKernelGGL(chain_lim_roles_kernel, dim3(groups), dim3(64 * kLimWaves), lim_roles_lds_bytes(max_lookahead), stream, args, args.params);
  return hipGetLastError();
}

// the compressor of `args.n_samples` steps for every stream, in -> out (may be the same buffer); needs comp_roles_serves()
hipError_t launch_chain_comp_roles(const LaunchArgs &args, bool sidechain, bool adaptive, hipStream_t stream) {
  const int groups = (args.n_streams + kLanes - 1) / kLanes;
  const dim3 grid(groups), block(64 * kCompWaves);
  const size_t lds = sizeof(CompLds);
#define AF_COMP_ROLES(SC, AD)                                                                                                  \
  do {                                                                                                                         \
    static bool attr_set = false;                                                                                              \
    if (!attr_set) {                                                                                                           \
      hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(chain_comp_roles_kernel<SC, AD>),                    \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
      if (err != hipSuccess) return err;                                                                                       \
      attr_set = true;                                                                                                         \
    }                                                                                                                          \
    hipLaunchKernelGGL((chain_comp_roles_kernel<SC, AD>), grid, block, lds, stream, args, args.params);                        \
  } while (0)
  if (sidechain) {
    if (adaptive) AF_COMP_ROLES(true, true);
    else AF_COMP_ROLES(true, false);
  } else {
    if (adaptive) AF_COMP_ROLES(false, true);
    else AF_COMP_ROLES(false, false);
  }
#undef AF_COMP_ROLES
  return hipGetLastError();
}

}  // namespace af
