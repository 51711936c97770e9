// af_stages.hip -- Kernel 4: the dynamics chain (compressor -> limiter -> true-peak limiter -> output detector) as a
// PIPELINE OF STAGE KERNELS.
//
// The token-ring kernel (af_ring_kernel.hip) keeps a 64-stream group on one CU: whatever the batch, a launch lasts as long
// as one workgroup needs for its streams' samples (~1000 cycles per sample step), 256 streams use 4 CUs and 4096 use 64.
// The chain is a string of short recurrences (a few dependent f64 operations per sample each) separated by feed-forward
// math (log10 / exp10 / sqrt / divisions, two 128-tap FIRs, a sliding maximum).  Here every recurrence is a kernel of its
// own -- one wave per 64-stream group, lane = stream, its state in REGISTERS for the whole window, nothing but the
// recurrence in its loop -- and every feed-forward piece is a wide elementwise stage over (sample, stream).  Time is cut
// into windows; launch step j runs stage k on window j - skew(k) for all stages at once (roles of two dispatches, picked
// by block index), so while stage k works on window w stage k+1 works on window w-1: a step lasts as long as the SLOWEST
// STAGE needs for a window, not the sum, the wide work spreads over the whole chip, and stream order between the steps is
// all the synchronisation there is.  Hand-over between stages is through rings in HBM (af_stages.h), ~0.3 KB per sample step per
// stream: nothing at a few hundred streams, the reason large batches stay on the token-ring kernel.
//
// What bounds a serial stage (tools/probe/valu_latency.hip): a lone wave issues a dependent vector instruction every
// ~8.3 cycles and an independent one every ~5, whatever its type, and may have 63 memory instructions in flight.  So a
// stage's loop holds nothing but its recurrence, and a lane's four consecutive samples are contiguous in the rings: one
// 16-byte load or store moves four steps (a quarter of the memory instructions of a row-per-step layout).
//
// Arithmetic: the expressions are those of the token-ring kernel, operation for operation (the build does not contract
// floating-point expressions), so the two kernels agree bit for bit; tests/test_gpu_stages.py holds them to that.
//
// Not built in this form (the host keeps such configurations on kernel 2): the de-esser, the front end
// without the suppressor, more than 16 EQ sections, presets that differ in which stages run, the time-major layout.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "af_deesser_math.h"
#include "af_dsp.h"
#include "af_eq_systolic_body.h"
#include "af_stages.h"

namespace af {
namespace {

constexpr int kQ = 4;          // steps per quad: a lane's kQ consecutive samples are contiguous in every ring
constexpr int kU = 16;         // steps per unrolled block of a serial stage (kU / kQ quads)
constexpr int kTileRows = 64;  // steps per workgroup of a tiled feed-forward stage (4 waves x 16 steps)
constexpr int kQuadRow = kLanes * kQ;

// ring addressing: element (sample n, lane l) of a group = quad (n >> 2) [modulo the ring], then lane, then n & 3
__device__ __forceinline__ int64_t qoff(int64_t q, int rows) { return (q & (int64_t)(rows / kQ - 1)) * kQuadRow; }
__device__ __forceinline__ int64_t eoff(int64_t n, int lane, int rows) { return qoff(n >> 2, rows) + lane * kQ + (n & 3); }

template <typename T>
struct Quad {
  T v[kQ];
};
__device__ __forceinline__ Quad<float> load_quad(const float *p) {
  const float4 x = *reinterpret_cast<const float4 *>(p);
  return Quad<float>{{x.x, x.y, x.z, x.w}};
}
__device__ __forceinline__ Quad<double> load_quad(const double *p) {
  const double2 a = *reinterpret_cast<const double2 *>(p), b = *reinterpret_cast<const double2 *>(p + 2);
  return Quad<double>{{a.x, a.y, b.x, b.y}};
}
__device__ __forceinline__ void store_quad(float *p, const float (&v)[kU], int k) {
  *reinterpret_cast<float4 *>(p) = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
}
__device__ __forceinline__ void store_quad(double *p, const double (&v)[kU], int k) {
  *reinterpret_cast<double2 *>(p) = make_double2(v[4 * k], v[4 * k + 1]);
  *reinterpret_cast<double2 *>(p + 2) = make_double2(v[4 * k + 2], v[4 * k + 3]);
}

// what every stage works out first
struct Who {
  int lane, g, s, sc;
  bool valid;
  int64_t NS;
};
__device__ __forceinline__ Who who(const StageArgs &a, int g) {
  Who w;
  w.lane = threadIdx.x & (kLanes - 1);
  w.g = g;
  w.s = g * kLanes + w.lane;
  w.valid = w.s < a.n_streams;
  w.sc = w.valid ? w.s : a.n_streams - 1;
  w.NS = a.n_streams;
  return w;
}
__device__ __forceinline__ const ChainParams &preset(const StageArgs &a, int g) {
  return a.params[a.group_preset ? a.group_preset[g] : 0];
}

// Input of a serial stage: three blocks of kU steps (four quads each) in registers; the loads of block i + 3 are issued
// when block i is done, into the buffer block i has just left (the loop is written out three times, so the buffers rotate
// by name: a rotation by copying costs four register moves per step, a third of a lean stage's instructions).  Slot u of
// a block is absolute sample 4 * qb + u (qb = the block's first quad).
template <typename T>
struct Ahead {
  T b0[kU], b1[kU], b2[kU];
  const T *base;  // ring of this group
  int rows, lane;
  int64_t shift;  // in quads: read sample n + 4 * shift where the stage is at sample n (a delay line)
  __device__ __forceinline__ void fill(T (&v)[kU], int64_t qb) {
#pragma unroll
    for (int k = 0; k < kU / kQ; ++k) {
      const Quad<T> x = load_quad(base + qoff(qb + shift + k, rows) + lane * kQ);
#pragma unroll
      for (int j = 0; j < kQ; ++j) v[kQ * k + j] = x.v[j];
    }
  }
  __device__ __forceinline__ void init(const T *ring, int g, int lane_, int rows_, int64_t qb, int64_t shift_ = 0) {
    base = ring + (int64_t)g * rows_ * kLanes;
    rows = rows_;
    lane = lane_;
    shift = shift_;
    fill(b0, qb);
    fill(b1, qb + kU / kQ);
    fill(b2, qb + 2 * (kU / kQ));
  }
  template <int I>
  __device__ __forceinline__ T (&buf())[kU] {
    if constexpr (I == 0) return b0;
    else if constexpr (I == 1) return b1;
    else return b2;
  }
  template <int I>
  __device__ __forceinline__ void refill(int64_t qb) {  // qb: the block buffer I has just served
    fill(buf<I>(), qb + 3 * (kU / kQ));
  }
};

// the serial stages' loop over blocks of kU steps: f(qb, buffer index)
template <typename F>
__device__ __forceinline__ void for_blocks(int64_t q_first, int64_t q_last, F f) {
  int64_t qb = q_first;
  while (true) {
    if (qb > q_last) break;
    f(qb, std::integral_constant<int, 0>{});
    qb += kU / kQ;
    if (qb > q_last) break;
    f(qb, std::integral_constant<int, 1>{});
    qb += kU / kQ;
    if (qb > q_last) break;
    f(qb, std::integral_constant<int, 2>{});
    qb += kU / kQ;
  }
}

// One block of kU slots of a serial stage.  The slots that belong to the window, [lo, hi), run as whole-block unrolled code
// with vector stores when all kU do and no control block ends inside; otherwise slot by slot with element stores (the
// window's first and last block, and the blocks a control block ends in).  `block_end` exists once, outside the unrolled code.
template <bool kBlocks, typename Step, typename StoreAll, typename StoreOne, typename BlockEnd>
__device__ __forceinline__ void run_block(int64_t qb, int64_t n0, int64_t n, int cb, int &in_block, Step step, StoreAll store_all,
                                          StoreOne store_one, BlockEnd block_end) {
  const int64_t base = qb * kQ;
  const int lo = n0 > base ? (int)(n0 - base) : 0;
  const int hi = (n0 + n - base) < kU ? (int)(n0 + n - base) : kU;
  int u0 = lo;
  while (u0 < hi) {
    int seg = hi - u0;
    if (kBlocks && cb - in_block < seg) seg = cb - in_block;
    if (seg == kU) {
#pragma unroll
      for (int u = 0; u < kU; ++u) step(u);
      store_all();
    } else {
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (u >= u0 && u < u0 + seg) {
          step(u);
          store_one(u);
        }
    }
    u0 += seg;
    if (kBlocks) {
      in_block += seg;
      if (in_block == cb || base + u0 == n0 + n) {
        block_end();
        in_block = 0;
      }
    }
  }
}

// an output ring of a serial stage: the block's kU results in registers, stored as four quads or slot by slot
template <typename T>
struct Out {
  T v[kU];
  T *base;
  int rows, lane;
  __device__ __forceinline__ void init(T *ring, int g, int lane_, int rows_) {
    base = ring + (int64_t)g * rows_ * kLanes;
    rows = rows_;
    lane = lane_;
  }
  __device__ __forceinline__ void store_all(int64_t qb) {
#pragma unroll
    for (int k = 0; k < kU / kQ; ++k) store_quad(base + qoff(qb + k, rows) + lane * kQ, v, k);
  }
  __device__ __forceinline__ void store_one(int64_t qb, int u) { base[eoff(qb * kQ + u, lane, rows)] = v[u]; }
};

// elementwise feed-forward stages: a workgroup takes kFfQuads quad-rows of a group, a thread element i of each
// (lane i >> 2, step i & 3): consecutive threads touch consecutive addresses
constexpr int kFfQuads = 8;
struct Elem {
  int64_t idx;   // offset inside a group's ring
  int64_t abs;   // absolute sample
  bool in;       // inside the window
};
__device__ __forceinline__ Elem ff_elem(const StageArgs &a, int64_t q, int i, int rows) {
  Elem e;
  e.abs = q * kQ + (i & 3);
  e.idx = qoff(q, rows) + i;
  e.in = e.abs >= a.n0 && e.abs < a.n0 + a.n;
  return e;
}

// ============================================================================================ block input statistics
// input_square_sum (f64, in sample order) and input_sample_peak of every control block (block_processor.rs:111-118) from
// the scrubbed input the EQ kernel leaves in the `xi` ring: two instructions on the recurrence, off the EQ's own loop
__device__ __forceinline__ void stage_in_body(const StageArgs &a, int bx, int /*by*/) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  Ahead<float> in;
  in.init(a.r.xi, w.g, w.lane, a.r.rows_f32, q_first);
  const int cb = P.control_block;
  BlockStats *stats = a.stats;
  double in_sq = 0.0;
  float in_peak = 0.0f;
  int in_block = 0;
  int64_t b = 0;
  auto block_end = [&]() {
    if (w.valid && stats) {
      stats[b * w.NS + w.s].input_square_sum = in_sq;
      stats[b * w.NS + w.s].input_sample_peak = in_peak;
    }
    in_sq = 0.0;
    in_peak = 0.0f;
    b += 1;
  };
  for_blocks(q_first, q_last, [&](int64_t qb, auto buf_tag) {
    constexpr int kBuf = decltype(buf_tag)::value;
    const float(&cur)[kU] = in.template buf<kBuf>();
    auto step = [&](int u) {
      const float v = cur[u];
      in_sq += (double)v * (double)v;
      in_peak = fmaxf(in_peak, fabsf(v));
    };
    run_block<true>(qb, n0, n, cb, in_block, step, [] {}, [](int) {}, block_end);
    in.template refill<kBuf>(qb);
  });
}

// ============================================================================================ compressor, serial part A
// side-chain high-pass + band / rms envelopes (compressor.rs:700-733; the token-ring kernel's token A), as two stages of
// about a dozen instructions per step each: (1) the high-pass, the low band's envelope and the presence signal, (2) the
// voiced, presence and rms envelopes
template <bool kSc>
__device__ __forceinline__ void stage_comp_a_body(const StageArgs &a, int bx, int /*by*/) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  const CompressorParams &cp = P.comp;
  __builtin_amdgcn_s_setprio(3);
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  Ahead<float> in;
  in.init(a.r.xe, w.g, w.lane, a.r.rows_f32, q_first);
  Out<double> o_d, o_low, o_pr;
  o_d.init(a.r.d, w.g, w.lane, a.r.rows_f64);
  if (kSc) {
    o_low.init(a.r.low_e, w.g, w.lane, a.r.rows_f64);
    o_pr.init(a.r.pr, w.g, w.lane, a.r.rows_f64);
  }
  double prev_in = a.st64[(int64_t)kCompScPrevIn * w.NS + w.sc], prev_out = a.st64[(int64_t)kCompScPrevOut * w.NS + w.sc];
  double low_env = a.st64[(int64_t)kCompLowEnv * w.NS + w.sc];
  // every parameter the loop reads is copied out first: read through the parameter pointer it would be re-loaded from
  // memory at every step (the loop's stores could alias it), each time behind a wait for ALL outstanding memory traffic
  const double kk = cp.band_env_coeff, sc_coeff = cp.sidechain_highpass_coeff;
  const double one_m_kk = 1.0 - kk;
  int dummy = 0;
  for_blocks(q_first, q_last, [&](int64_t qb, auto buf_tag) {
    constexpr int kBuf = decltype(buf_tag)::value;
    const float(&cur)[kU] = in.template buf<kBuf>();
    auto step = [&](int u) {
      const double xin = (double)cur[u];
      if (kSc) {
        const double dd = sc_coeff * (prev_out + xin - prev_in);
        prev_in = xin;
        prev_out = dd;
        const double low = xin - dd;
        const double presence = 0.65 * dd + 0.35 * (dd - low);
        low_env = kk * low_env + one_m_kk * low * low;
        o_d.v[u] = dd;
        o_pr.v[u] = presence;
        o_low.v[u] = low_env;
      } else {
        o_d.v[u] = xin;
      }
    };
    auto store_all = [&]() {
      o_d.store_all(qb);
      if (kSc) {
        o_low.store_all(qb);
        o_pr.store_all(qb);
      }
    };
    auto store_one = [&](int u) {
      o_d.store_one(qb, u);
      if (kSc) {
        o_low.store_one(qb, u);
        o_pr.store_one(qb, u);
      }
    };
    run_block<false>(qb, n0, n, 0, dummy, step, store_all, store_one, [] {});
    in.template refill<kBuf>(qb);
  });
  if (w.valid && kSc) {
    a.st64[(int64_t)kCompScPrevIn * w.NS + w.s] = prev_in;
    a.st64[(int64_t)kCompScPrevOut * w.NS + w.s] = prev_out;
    a.st64[(int64_t)kCompLowEnv * w.NS + w.s] = low_env;
  }
}

template <bool kSc>
__device__ __forceinline__ void stage_comp_a2_body(const StageArgs &a, int bx, int /*by*/) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  const CompressorParams &cp = P.comp;
  __builtin_amdgcn_s_setprio(3);
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  Ahead<double> in_d, in_pr;
  in_d.init(a.r.d, w.g, w.lane, a.r.rows_f64, q_first);
  if (kSc) in_pr.init(a.r.pr, w.g, w.lane, a.r.rows_f64, q_first);
  Out<double> o_voiced, o_pres, o_rms;
  o_rms.init(a.r.rms_e, w.g, w.lane, a.r.rows_f64);
  if (kSc) {
    o_voiced.init(a.r.voiced_e, w.g, w.lane, a.r.rows_f64);
    o_pres.init(a.r.pres_e, w.g, w.lane, a.r.rows_f64);
  }
  double rms_env = a.st64[(int64_t)kCompRmsEnvSq * w.NS + w.sc];
  double voiced_env = a.st64[(int64_t)kCompVoicedEnv * w.NS + w.sc], presence_env = a.st64[(int64_t)kCompPresenceEnv * w.NS + w.sc];
  const double kk = cp.band_env_coeff, rms_coeff = cp.rms_coeff;
  const double one_m_kk = 1.0 - kk, one_m_rms = 1.0 - rms_coeff;
  int dummy = 0;
  for_blocks(q_first, q_last, [&](int64_t qb, auto buf_tag) {
    constexpr int kBuf = decltype(buf_tag)::value;
    const double(&cur_d)[kU] = in_d.template buf<kBuf>();
    const double(&cur_pr)[kU] = in_pr.template buf<kBuf>();
    auto step = [&](int u) {
      const double dd = cur_d[u];
      if (kSc) {
        const double presence = cur_pr[u];
        voiced_env = kk * voiced_env + one_m_kk * dd * dd;
        presence_env = kk * presence_env + one_m_kk * presence * presence;
        o_voiced.v[u] = voiced_env;
        o_pres.v[u] = presence_env;
      }
      rms_env = rms_coeff * rms_env + one_m_rms * (dd * dd);
      o_rms.v[u] = rms_env;
    };
    auto store_all = [&]() {
      o_rms.store_all(qb);
      if (kSc) {
        o_voiced.store_all(qb);
        o_pres.store_all(qb);
      }
    };
    auto store_one = [&](int u) {
      o_rms.store_one(qb, u);
      if (kSc) {
        o_voiced.store_one(qb, u);
        o_pres.store_one(qb, u);
      }
    };
    run_block<false>(qb, n0, n, 0, dummy, step, store_all, store_one, [] {});
    in_d.template refill<kBuf>(qb);
    if (kSc) in_pr.template refill<kBuf>(qb);
  });
  if (w.valid) {
    a.st64[(int64_t)kCompRmsEnvSq * w.NS + w.s] = rms_env;
    if (kSc) {
      a.st64[(int64_t)kCompVoicedEnv * w.NS + w.s] = voiced_env;
      a.st64[(int64_t)kCompPresenceEnv * w.NS + w.s] = presence_env;
    }
  }
}

// ============================================================================================ feed-forward 1
// detector weight, instantaneous peak and RMS levels in dB (update_sidechain_band_metrics, compressor.rs:438-449)
__device__ __forceinline__ void stage_f1_body(const StageArgs &a, int bx, int by) {
  const int g = by;
  const ChainParams &P = preset(a, g);
  const CompressorParams cp = P.comp;  // by value: fields read through the pointer would be re-loaded after every store
  const int R = a.r.rows_f64;
  const int64_t gb = (int64_t)g * R * kLanes;
  const int i = threadIdx.x, lane = i >> 2;
  const int s = g * kLanes + lane;
  const int64_t NS = a.n_streams;
  const int64_t q0 = (a.n0 >> 2) + (int64_t)bx * kFfQuads;
  double i_low[kFfQuads], i_voiced[kFfQuads], i_pres[kFfQuads], i_rms[kFfQuads], i_d[kFfQuads];  // (loaded ahead: see stage F5)
#pragma unroll
  for (int k = 0; k < kFfQuads; ++k) {
    const int64_t row = gb + qoff(q0 + k, R) + i;
    if (cp.sidechain_highpass_enabled) {
      i_low[k] = a.r.low_e[row];
      i_voiced[k] = a.r.voiced_e[row];
      i_pres[k] = a.r.pres_e[row];
    }
    i_rms[k] = a.r.rms_e[row];
    i_d[k] = a.r.d[row];
  }
#pragma unroll
  for (int k = 0; k < kFfQuads; ++k) {
    const Elem e = ff_elem(a, q0 + k, i, R);
    if (!e.in) continue;
    const int64_t row = gb + e.idx;
    double weight_db = kDetectorUnitWeight, plosive_last = 0.0;  // (the w_db / rms_db rings: dB in the literal build, linear otherwise)
    if (cp.sidechain_highpass_enabled) {
      const double low_rms = sqrt(i_low[k]);
      const double voiced_rms = fmax(sqrt(i_voiced[k]), 1e-8);
      const double presence_rms = sqrt(i_pres[k]);
      const double plosive = dclamp(low_rms / voiced_rms, 0.0, 32.0);
      plosive_last = plosive;
      const double plosive_amount = dclamp(div_known(plosive - 1.25, 3.75, 1.0 / 3.75), 0.0, 1.0);
      const double plosive_penalty = 1.0 - plosive_amount * (1.0 - 0.35);
      const double presence_ratio = dclamp(presence_rms / voiced_rms, 0.0, 4.0);
      const double presence_weight = 1.0 + 0.18 * dclamp(presence_ratio - 0.75, 0.0, 1.0);
      weight_db = detector_weight(dclamp(plosive_penalty * presence_weight, 0.35, 1.15));
    }
    a.r.w_db[row] = weight_db;
    a.r.ipk_db[row] = lin2db(fabs(i_d[k]), 1e-10);
    a.r.rms_db[row] = detector_rms_level(i_rms[k]);
    if (e.abs == a.n0 + a.n - 1 && s < a.n_streams) a.st64[(int64_t)kCompPlosive * NS + s] = plosive_last;  // diagnostic state only
  }
}

// ============================================================================================ compressor, serial part C
// log-domain peak envelope (compressor.rs:735-742)
__device__ __forceinline__ void stage_comp_c_body(const StageArgs &a, int bx, int /*by*/) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  const CompressorParams &cp = P.comp;
  __builtin_amdgcn_s_setprio(3);
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  Ahead<double> in;
  in.init(a.r.ipk_db, w.g, w.lane, a.r.rows_f64, q_first);
  Out<double> o;
  o.init(a.r.peak_db, w.g, w.lane, a.r.rows_f64);
  double pe = a.st64[(int64_t)kCompPeakEnvDb * w.NS + w.sc];
  const double attack_coeff = cp.attack_coeff, detector_release_coeff = cp.detector_release_coeff;
  // both candidates of the one-pole step are formed before the comparison picks one: the same operations on the same
  // values as `pk = v > pe ? attack : release; pe = pk * pe + (1 - pk) * v`, but only mul -> add -> select sits on the
  // recurrence's critical path (the products with v do not depend on pe)
  const double one_m_attack = 1.0 - attack_coeff, one_m_release = 1.0 - detector_release_coeff;
  int dummy = 0;
  for_blocks(q_first, q_last, [&](int64_t qb, auto buf_tag) {
    constexpr int kBuf = decltype(buf_tag)::value;
    const double(&cur)[kU] = in.template buf<kBuf>();
    auto step = [&](int u) {
      const double v = cur[u];
      const double rising = attack_coeff * pe + one_m_attack * v;
      const double falling = detector_release_coeff * pe + one_m_release * v;
      pe = v > pe ? rising : falling;
      o.v[u] = pe;
    };
    run_block<false>(qb, n0, n, 0, dummy, step, [&] { o.store_all(qb); }, [&](int u) { o.store_one(qb, u); }, [] {});
    in.template refill<kBuf>(qb);
  });
  if (w.valid) a.st64[(int64_t)kCompPeakEnvDb * w.NS + w.s] = pe;
}

// ============================================================================================ feed-forward 2
// blended detector level -> static gain-reduction target (compressor.rs:744-750,657-678)
__device__ __forceinline__ void stage_f2_body(const StageArgs &a, int bx, int by) {
  const int g = by;
  const ChainParams &P = preset(a, g);
  const CompressorParams cp = P.comp;
  const int R = a.r.rows_f64;
  const int64_t gb = (int64_t)g * R * kLanes;
  const int i = threadIdx.x;
  const int64_t q0 = (a.n0 >> 2) + (int64_t)bx * kFfQuads;
  double pk[kFfQuads], rm[kFfQuads], wd[kFfQuads];  // (loaded ahead of the arithmetic: see stage F5)
#pragma unroll
  for (int k = 0; k < kFfQuads; ++k) {
    const int64_t row = gb + qoff(q0 + k, R) + i;
    pk[k] = a.r.peak_db[row];
    rm[k] = a.r.rms_db[row];
    wd[k] = a.r.w_db[row];
  }
#pragma unroll
  for (int k = 0; k < kFfQuads; ++k) {
    const Elem e = ff_elem(a, q0 + k, i, R);
    if (!e.in) continue;
    const int64_t row = gb + e.idx;
    a.r.target[row] = comp_gain_reduction(cp, detector_db(pk[k], rm[k], wd[k]));
  }
}

// ============================================================================================ compressor, serial part E
// release-time meter + gain-reduction smoothing, makeup gain per control block (compressor.rs:452-505,604-617,752-764)
template <bool kAdaptive, bool kAuto>
__device__ __forceinline__ void stage_comp_e_body(const StageArgs &a, int bx, int /*by*/) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  const CompressorParams &cp = P.comp;
  __builtin_amdgcn_s_setprio(3);
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  Ahead<double> in;
  in.init(a.r.target, w.g, w.lane, a.r.rows_f64, q_first);
  Out<double> o, o_fast, o_slow;
  o.init(a.r.gr, w.g, w.lane, a.r.rows_f64);
  if (kAdaptive) {  // what the release-time meter (stages FR + Rel) reads: the envelopes as each step found them
    o_fast.init(a.r.fast_r, w.g, w.lane, a.r.rows_f64);
    o_slow.init(a.r.slow_r, w.g, w.lane, a.r.rows_f64);
  }
  double gr = a.st64[(int64_t)kCompGr * w.NS + w.sc], fast = a.st64[(int64_t)kCompFastEnv * w.NS + w.sc];
  double slow = a.st64[(int64_t)kCompSlowEnv * w.NS + w.sc];
  double cur_ms = a.st64[(int64_t)kCompCurReleaseMs * w.NS + w.sc], tgt_ms = a.st64[(int64_t)kCompTargetReleaseMs * w.NS + w.sc];
  const double rel_coeff = a.st64[(int64_t)kCompReleaseCoeff * w.NS + w.sc];
  double sm = a.st64[(int64_t)kCompSmoothedMakeup * w.NS + w.sc];
  double makeup_lin = db2lin(sm);
  const int cb = P.control_block;
  const double base_release_ms = cp.base_release_ms, release_smoothing_coeff = cp.release_smoothing_coeff;
  const double attack_coeff = cp.attack_coeff, fast_release_coeff = cp.fast_release_coeff, slow_charge_coeff = cp.slow_charge_coeff;
  const double slow_release_coeff = cp.slow_release_coeff, makeup_smoothing_coeff = cp.makeup_smoothing_coeff, makeup_gain_db = cp.makeup_gain_db;
  double *mk = a.mk;
  BlockStats *stats = a.stats;
  int in_block = 0;
  int64_t b = 0;
  if (w.valid && !kAuto) mk[w.s] = makeup_lin;  // the gain in force during the window's first block
  auto block_end = [&]() {  // wave-uniform: a control block (or the window) ends after the step just made
    const int blk_len = in_block;
    if (w.valid && stats) stats[b * w.NS + w.s].compressor_gr_db = (float)gr;
    if (kAuto) {  // the makeup gain belongs to the makeup stage
      b += 1;
      return;
    }
    // update_auto_makeup_gain with auto-makeup off (compressor.rs:604-617)
    const double makeup_coeff = pow(makeup_smoothing_coeff, (double)(blk_len < 1 ? 1 : blk_len));
    const double tgt = makeup_gain_db;
    if (fabs(tgt - sm) > 0.1) {
      sm = makeup_coeff * sm + (1.0 - makeup_coeff) * tgt;
    } else {
      sm = tgt;
    }
    makeup_lin = db2lin(sm);
    if (w.valid && stats) stats[b * w.NS + w.s].makeup_gain_db = (float)sm;
    b += 1;
    if (w.valid) mk[b * w.NS + w.s] = makeup_lin;  // ... and during the next one
  };
  const double one_m_attack = 1.0 - attack_coeff, one_m_rel = 1.0 - rel_coeff, one_m_fast = 1.0 - fast_release_coeff;
  const double one_m_charge = 1.0 - slow_charge_coeff, one_m_smooth = 1.0 - release_smoothing_coeff;
  for_blocks(q_first, q_last, [&](int64_t qb, auto buf_tag) {
    constexpr int kBuf = decltype(buf_tag)::value;
    const double(&cur)[kU] = in.template buf<kBuf>();
    auto step = [&](int u) {
      const double tg = cur[u];
      if (kAdaptive) {
        // update_adaptive_release_time_meter (compressor.rs:452-466) feeds nothing in this loop: it runs as stages FR + Rel
        o_fast.v[u] = fast;
        o_slow.v[u] = slow;
      } else {
        tgt_ms = base_release_ms;
        const double smoothed = release_smoothing_coeff * cur_ms + one_m_smooth * tgt_ms;
        cur_ms = fabs(tgt_ms - cur_ms) > 1.0 ? smoothed : tgt_ms;
      }
      // (both candidates of every one-pole step are formed before the comparison picks one: same operations, same values)
      if (!kAdaptive) {
        const double rising = attack_coeff * gr + one_m_attack * tg;
        const double falling = rel_coeff * gr + one_m_rel * tg;
        gr = tg > gr ? rising : falling;
        fast = gr;
        slow = 0.0;
      } else {
        const double rising = attack_coeff * gr + one_m_attack * tg;
        const double falling = fast_release_coeff * fast + one_m_fast * tg;
        fast = tg > gr ? rising : falling;
        const double charged = slow_charge_coeff * slow + one_m_charge * tg;
        const double released = slow * slow_release_coeff;
        slow = tg > 3.0 ? charged : released;
        gr = fmax(fast, slow);
      }
      o.v[u] = gr;
    };
    run_block<true>(
        qb, n0, n, cb, in_block, step,
        [&] {
          o.store_all(qb);
          if (kAdaptive) {
            o_fast.store_all(qb);
            o_slow.store_all(qb);
          }
        },
        [&](int u) {
          o.store_one(qb, u);
          if (kAdaptive) {
            o_fast.store_one(qb, u);
            o_slow.store_one(qb, u);
          }
        },
        block_end);
    in.template refill<kBuf>(qb);
  });
  if (w.valid) {
    a.st64[(int64_t)kCompGr * w.NS + w.s] = gr;
    a.st64[(int64_t)kCompFastEnv * w.NS + w.s] = fast;
    a.st64[(int64_t)kCompSlowEnv * w.NS + w.s] = slow;
    if (!kAdaptive) {
      a.st64[(int64_t)kCompCurReleaseMs * w.NS + w.s] = cur_ms;
      a.st64[(int64_t)kCompTargetReleaseMs * w.NS + w.s] = tgt_ms;
    }
    if (!kAuto) a.st64[(int64_t)kCompSmoothedMakeup * w.NS + w.s] = sm;
  }
}

// release-time meter, adaptive release only (compressor.rs:452-466,752-761): the target release time is a function of the
// envelopes each step found (wide stage FR), its smoothing a recurrence of five instructions (serial stage Rel)
__device__ __forceinline__ void stage_fr_body(const StageArgs &a, int bx, int by) {
  const int g = by;
  const int R = a.r.rows_f64;
  const int64_t gb = (int64_t)g * R * kLanes;
  const int i = threadIdx.x;
  const int64_t q0 = (a.n0 >> 2) + (int64_t)bx * kFfQuads;
  for (int k = 0; k < kFfQuads; ++k) {
    const Elem e = ff_elem(a, q0 + k, i, R);
    if (!e.in) continue;
    const int64_t row = gb + e.idx;
    const double fast = a.r.fast_r[row], slow = a.r.slow_r[row];
    const double sustained = dclamp(div_known(slow, 6.0, 1.0 / 6.0), 0.0, 1.0);
    const double transient_bias = dclamp(div_known(fast - slow, 7.0, 1.0 / 7.0), 0.0, 1.0);
    const double syllabic = dclamp(sustained * sustained * (1.0 - 0.35 * transient_bias), 0.0, 1.0);
    a.r.tgt_ms[row] = 50.0 + syllabic * (400.0 - 50.0);
  }
}

__device__ __forceinline__ void stage_rel_body(const StageArgs &a, int bx, int /*by*/) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  const CompressorParams &cp = P.comp;
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  Ahead<double> in;
  in.init(a.r.tgt_ms, w.g, w.lane, a.r.rows_f64, q_first);
  double cur_ms = a.st64[(int64_t)kCompCurReleaseMs * w.NS + w.sc], tgt_ms = a.st64[(int64_t)kCompTargetReleaseMs * w.NS + w.sc];
  const double release_smoothing_coeff = cp.release_smoothing_coeff, sample_rate = cp.sample_rate;
  const double one_m_smooth = 1.0 - release_smoothing_coeff;
  int dummy = 0;
  for_blocks(q_first, q_last, [&](int64_t qb, auto buf_tag) {
    constexpr int kBuf = decltype(buf_tag)::value;
    const double(&cur)[kU] = in.template buf<kBuf>();
    auto step = [&](int u) {
      tgt_ms = cur[u];
      const double smoothed = release_smoothing_coeff * cur_ms + one_m_smooth * tgt_ms;
      cur_ms = fabs(tgt_ms - cur_ms) > 1.0 ? smoothed : tgt_ms;
    };
    run_block<false>(qb, n0, n, 0, dummy, step, [] {}, [](int) {}, [] {});
    in.template refill<kBuf>(qb);
  });
  if (w.valid) {
    a.st64[(int64_t)kCompCurReleaseMs * w.NS + w.s] = cur_ms;
    a.st64[(int64_t)kCompTargetReleaseMs * w.NS + w.s] = tgt_ms;
    const double tau = fmax(cur_ms, 0.001) / 1000.0;  // compressor.rs:760-761
    a.st64[(int64_t)kCompReleaseCoeff * w.NS + w.s] = exp(-1.0 / (tau * sample_rate));
  }
}

// ============================================================================================ auto-makeup (compressor.rs:528-653)
// The controller needs the RMS of each whole control block of the compressor's INPUT before the block's first sample
// (compressor.rs:710).  The token-ring kernel runs a launch of its own for that; here it is one more serial stage, seven
// launch steps ahead of the stage that uses it.
__device__ __forceinline__ void stage_pow_body(const StageArgs &a, int bx, int /*by*/) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  Ahead<float> in;
  in.init(a.r.xe, w.g, w.lane, a.r.rows_f32, q_first);
  const int cb = P.control_block;
  double *bp = a.bp;
  double sq = 0.0;
  int in_block = 0;
  int64_t b = 0;
  auto block_end = [&]() {
    if (w.valid) bp[b * w.NS + w.s] = sq;
    sq = 0.0;
    b += 1;
  };
  for_blocks(q_first, q_last, [&](int64_t qb, auto buf_tag) {
    constexpr int kBuf = decltype(buf_tag)::value;
    const float(&cur)[kU] = in.template buf<kBuf>();
    auto step = [&](int u) {
      const float o = cur[u];
      if (finite_f32(o)) sq += (double)o * (double)o;
    };
    run_block<true>(qb, n0, n, cb, in_block, step, [] {}, [](int) {}, block_end);
    in.template refill<kBuf>(qb);
  });
}

__device__ __forceinline__ void stage_f3a_body(const StageArgs &a, int bx, int by) {
  const int g = by;
  const int R = a.r.rows_f64;
  const int64_t gb = (int64_t)g * R * kLanes;
  const int i = threadIdx.x;
  const int64_t q0 = (a.n0 >> 2) + (int64_t)bx * kFfQuads;
  for (int k = 0; k < kFfQuads; ++k) {
    const Elem e = ff_elem(a, q0 + k, i, R);
    if (!e.in) continue;
    a.r.glin[gb + e.idx] = db2lin(-a.r.gr[gb + e.idx]);
  }
}

// makeup gain of the block, K-weighted momentary loudness of what leaves the compressor (fed blocks only), the controller
// at block end -- the token-ring kernel's makeup token, operation for operation
__device__ __forceinline__ void stage_makeup_body(const StageArgs &a, int bx, int /*by*/) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  const CompressorParams cp = P.comp;  // by value (see the serial stages above)
  __builtin_amdgcn_s_setprio(3);
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  Ahead<float> in_x;
  Ahead<double> in_g;
  in_x.init(a.r.xe, w.g, w.lane, a.r.rows_f32, q_first);
  in_g.init(a.r.glin, w.g, w.lane, a.r.rows_f64, q_first);
  Out<float> o;
  o.init(a.r.xc, w.g, w.lane, a.r.rows_f32);
  const int cb = P.control_block;
  const int mbase = kF64Fixed + 4 * P.n_eq_sections;
  const bool meter = cp.meter_slots > 0;
  double *st64 = a.st64;
  BlockStats *stats = a.stats;
  const double *bp = a.bp, *vad = a.vad;
  double sm = st64[(int64_t)kCompSmoothedMakeup * w.NS + w.sc];
  double mk = db2lin(sm);
  double v1 = meter ? st64[(int64_t)(mbase + kMeterV1) * w.NS + w.sc] : 0.0, v2 = meter ? st64[(int64_t)(mbase + kMeterV2) * w.NS + w.sc] : 0.0;
  double v3 = meter ? st64[(int64_t)(mbase + kMeterV3) * w.NS + w.sc] : 0.0, v4 = meter ? st64[(int64_t)(mbase + kMeterV4) * w.NS + w.sc] : 0.0;
  double score_state = st64[(int64_t)kCompActivityScore * w.NS + w.sc], relst_state = st64[(int64_t)kCompActivityReliability * w.NS + w.sc];
  double lufs = st64[(int64_t)kCompCurrentLufs * w.NS + w.sc];
  double acc = 0.0, act = 0.0, rel = 0.0;
  bool fed = false;
  int in_block = 0;
  int64_t b = 0;
  auto block_start = [&]() {
    // estimate_auto_makeup_activity(block_rms_db(buffer), evidence), compressor.rs:528-596,710
    const int64_t left = n - b * cb;
    const int blk_len = (int)(left < cb ? left : cb);
    const double power = bp[b * w.NS + w.sc] / (double)blk_len;
    const double brms_db = lin2db(sqrt(power), 1e-10);
    double absolute = 0.0;
    if (brms_db >= -55.0 && brms_db <= -6.0)
      absolute = fmin(dclamp(div_known(brms_db + 55.0, 12.0, 1.0 / 12.0), 0.0, 1.0), dclamp(div_known(-6.0 - brms_db, 6.0, 1.0 / 6.0), 0.0, 1.0));
    act = absolute;
    rel = 1.0;
    if (cp.has_evidence) {
      double vad_rel = cp.vad_reliability;
      double vad_p = vad ? vad[b * w.NS + w.sc] : 0.0;
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
    fed = act > 0.20 && rel >= 0.35 && cp.meter_slots > 0;  // compressor.rs:714-720
    acc = 0.0;
  };
  auto block_end = [&]() {
    const int blk_len = in_block;
    if (fed) {
      const double tiny = 2.2250738585072014e-308;
      if (fabs(v1) < tiny) v1 = 0.0;
      if (fabs(v2) < tiny) v2 = 0.0;
      if (fabs(v3) < tiny) v3 = 0.0;
      if (fabs(v4) < tiny) v4 = 0.0;
      // 400 ms window = the last meter_slots fed blocks (block energies instead of 19 200 samples)
      const int written = (int)st64[(int64_t)(mbase + kMeterPos) * w.NS + w.sc];
      const int pos = written + 1 == cp.meter_slots ? 0 : written + 1;
      if (w.valid) {
        st64[(int64_t)(mbase + kMeterRing + written) * w.NS + w.s] = acc;
        st64[(int64_t)(mbase + kMeterPos) * w.NS + w.s] = (double)pos;
      }
      double sum = 0.0;
      for (int m = 0; m < cp.meter_slots; ++m) {
        const double e = m == written ? acc : st64[(int64_t)(mbase + kMeterRing + m) * w.NS + w.sc];
        sum += e;
      }
      const double energy = sum / cp.meter_frames;
      lufs = energy <= 0.0 ? -HUGE_VAL : (double)(float)(10.0 * (log(energy) / log(10.0)) - 0.691);
    }
    // update_auto_makeup_gain, compressor.rs:598-653
    const bool whole = blk_len == cb;
    const double elapsed = (double)(blk_len < 1 ? 1 : blk_len);
    const double mc = whole ? cp.makeup_pow_cb : pow(cp.makeup_smoothing_coeff, elapsed);
    const double rc2 = whole ? cp.relax_pow_cb : pow(cp.makeup_silence_relax_coeff, elapsed);
    const double ac = whole ? cp.activity_pow_cb : pow(cp.speech_activity_smoothing_coeff, elapsed);
    const double score = ac * score_state + (1.0 - ac) * dclamp(act, 0.0, 1.0);
    const double relst = dclamp(rel, 0.0, 1.0);
    score_state = score;
    relst_state = relst;
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
    mk = db2lin(sm);
    if (w.valid && stats) {
      BlockStats &row = stats[b * w.NS + w.s];
      row.makeup_gain_db = (float)sm;
      row.makeup_activity = (float)score;
      row.makeup_reliability = (float)relst;
    }
    b += 1;
    if (b * cb < n) block_start();
  };
  if (n > 0) block_start();
  for_blocks(q_first, q_last, [&](int64_t qb, auto buf_tag) {
    constexpr int kBuf = decltype(buf_tag)::value;
    const float(&cur_x)[kU] = in_x.template buf<kBuf>();
    const double(&cur_g)[kU] = in_g.template buf<kBuf>();
    auto step = [&](int u) {
      const float x = (float)((double)cur_x[u] * (cur_g[u] * mk));
      if (fed) {  // K-weighting, one 4th-order direct-form section (loudness.rs:119-127 over ebur128)
        const double v0 = (double)x - cp.kw_a[1] * v1 - cp.kw_a[2] * v2 - cp.kw_a[3] * v3 - cp.kw_a[4] * v4;
        const double y = cp.kw_b[0] * v0 + cp.kw_b[1] * v1 + cp.kw_b[2] * v2 + cp.kw_b[3] * v3 + cp.kw_b[4] * v4;
        v4 = v3;
        v3 = v2;
        v2 = v1;
        v1 = v0;
        acc += y * y;
      }
      o.v[u] = x;
    };
    run_block<true>(qb, n0, n, cb, in_block, step, [&] { o.store_all(qb); }, [&](int u) { o.store_one(qb, u); }, block_end);
    in_x.template refill<kBuf>(qb);
    in_g.template refill<kBuf>(qb);
  });
  if (w.valid) {
    st64[(int64_t)kCompSmoothedMakeup * w.NS + w.s] = sm;
    st64[(int64_t)kCompActivityScore * w.NS + w.s] = score_state;
    st64[(int64_t)kCompActivityReliability * w.NS + w.s] = relst_state;
    st64[(int64_t)kCompCurrentLufs * w.NS + w.s] = lufs;
    if (meter) {
      st64[(int64_t)(mbase + kMeterV1) * w.NS + w.s] = v1;
      st64[(int64_t)(mbase + kMeterV2) * w.NS + w.s] = v2;
      st64[(int64_t)(mbase + kMeterV3) * w.NS + w.s] = v3;
      st64[(int64_t)(mbase + kMeterV4) * w.NS + w.s] = v4;
    }
  }
}

// ============================================================================================ feed-forward 3
// apply gain (compressor.rs:771-773)
__device__ __forceinline__ void stage_f3_body(const StageArgs &a, int bx, int by) {
  const int g = by;
  const ChainParams &P = preset(a, g);
  const int R = a.r.rows_f64, R32 = a.r.rows_f32;
  const int64_t gb = (int64_t)g * R * kLanes, gb32 = (int64_t)g * R32 * kLanes;
  const int i = threadIdx.x, lane = i >> 2;
  const int s = g * kLanes + lane;
  const int sc = s < a.n_streams ? s : a.n_streams - 1;
  const int64_t NS = a.n_streams;
  const int cb = P.control_block;
  const int64_t q0 = (a.n0 >> 2) + (int64_t)bx * kFfQuads;
  double i_gr[kFfQuads];  // (loaded ahead of the arithmetic: see stage F5)
  float i_x[kFfQuads];
#pragma unroll
  for (int k = 0; k < kFfQuads; ++k) {
    i_gr[k] = a.r.gr[gb + qoff(q0 + k, R) + i];
    i_x[k] = a.r.xe[gb32 + qoff(q0 + k, R32) + i];
  }
#pragma unroll
  for (int k = 0; k < kFfQuads; ++k) {
    const Elem e = ff_elem(a, q0 + k, i, R);
    if (!e.in) continue;
    const double makeup_lin = a.mk[((e.abs - a.n0) / cb) * NS + sc];
    const int64_t row = gb32 + qoff(q0 + k, R32) + i;
    a.r.xc[row] = (float)((double)i_x[k] * (db2lin(-i_gr[k]) * makeup_lin));
  }
}

// ============================================================================================ feed-forward 4
// lookahead limiter, the part without memory (limiter.rs:246-270): the maximum of |x| over the last W = lookahead + 1
// samples, from suffix maxima of the previous W-aligned block and the running prefix maximum of the current one (the
// maximum is exact whatever the grouping), and the gain it asks for.  One wave per (block that meets the window, group).
__device__ __forceinline__ void stage_f4_body(const StageArgs &a, const float *xin_ring, int bx, int by) {
  const int g = by;
  const Who w = who(a, g);
  const ChainParams &P = preset(a, g);
  const int W = P.lim.lookahead_samples + 1;
  const double ceil_lin = P.lim.ceiling_linear;
  const int R32 = a.r.rows_f32, R = a.r.rows_f64;
  const float *x = xin_ring + (int64_t)g * R32 * kLanes;
  float *sfx = a.r.sfx + (int64_t)g * R32 * kLanes;
  double *tg = a.r.tg + (int64_t)g * R * kLanes;
  const int lane = w.lane;
  const int64_t n0 = a.n0, n_end = a.n0 + a.n;
  // blocks are aligned to absolute multiples of W
  const int64_t B = n0 / W + bx;
  const int64_t b0 = B * W;
  if (b0 >= n_end) return;
  // suffix maxima of block B - 1, eight loads at a time
  {
    float m = 0.0f;
    int j = W - 1;
    for (; j >= 7; j -= 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = fabsf(x[eoff(b0 - W + j - u, lane, R32)]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        m = fmaxf(m, v[u]);
        sfx[eoff(b0 - W + j - u, lane, R32)] = m;
      }
    }
    for (; j >= 0; --j) {
      m = fmaxf(m, fabsf(x[eoff(b0 - W + j, lane, R32)]));
      sfx[eoff(b0 - W + j, lane, R32)] = m;
    }
  }
  // forward over block B: running prefix maximum, output for the samples that belong to this window
  float prefix = 0.0f;
  const int64_t end = (b0 + W < n_end) ? b0 + W : n_end;
  int64_t n = b0;
  for (; n + 8 <= end; n += 8) {
    float v[8], sf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      v[u] = fabsf(x[eoff(n + u, lane, R32)]);
      const int64_t j = n + u - b0;
      sf[u] = (j + 1 < W) ? sfx[eoff(b0 - W + j + 1, lane, R32)] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      prefix = (n + u == b0) ? v[u] : fmaxf(prefix, v[u]);
      if (n + u >= n0) {
        const double peak = (double)fmaxf(sf[u], prefix);
        tg[eoff(n + u, lane, R)] = peak > ceil_lin ? ceil_lin / peak : 1.0;
      }
    }
  }
  for (; n < end; ++n) {
    const float ax = fabsf(x[eoff(n, lane, R32)]);
    const int64_t j = n - b0;
    const float sf = (j + 1 < W) ? sfx[eoff(b0 - W + j + 1, lane, R32)] : 0.0f;
    prefix = (n == b0) ? ax : fmaxf(prefix, ax);
    if (n >= n0) {
      const double peak = (double)fmaxf(sf, prefix);
      tg[eoff(n, lane, R)] = peak > ceil_lin ? ceil_lin / peak : 1.0;
    }
  }
}

// ============================================================================================ limiter, serial part
// gain smoothing (limiter.rs:271-284)
__device__ __forceinline__ void stage_lim_body(const StageArgs &a, int bx, int /*by*/) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  __builtin_amdgcn_s_setprio(3);
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  Ahead<double> in;
  in.init(a.r.tg, w.g, w.lane, a.r.rows_f64, q_first);
  Out<double> o;
  o.init(a.r.g, w.g, w.lane, a.r.rows_f64);
  const double rc = P.lim.release_coeff;
  double g = a.st64[(int64_t)kLimGain * w.NS + w.sc];
  double gmin = 1.0;
  const int cb = P.control_block;
  BlockStats *stats = a.stats;
  int in_block = 0;
  int64_t b = 0;
  auto block_end = [&]() {
    if (w.valid && stats) stats[b * w.NS + w.s].limiter_peak_gr_db = gmin < 1.0 ? (float)(-lin2db(gmin, 1e-10)) : 0.0f;
    gmin = 1.0;
    b += 1;
  };
  const double one_m_rc = 1.0 - rc;
  for_blocks(q_first, q_last, [&](int64_t qb, auto buf_tag) {
    constexpr int kBuf = decltype(buf_tag)::value;
    const double(&cur)[kU] = in.template buf<kBuf>();
    auto step = [&](int u) {
      const double tg = cur[u];
      const double released = rc * g + one_m_rc * tg;
      g = tg < g ? tg : released;
      gmin = fmin(gmin, g);
      o.v[u] = g;
    };
    run_block<true>(qb, n0, n, cb, in_block, step, [&] { o.store_all(qb); }, [&](int u) { o.store_one(qb, u); }, block_end);
    in.template refill<kBuf>(qb);
  });
  if (w.valid) a.st64[(int64_t)kLimGain * w.NS + w.s] = g;
}

// ============================================================================================ feed-forward 5
// limiter output (limiter.rs:278-284), input-side 4x true peak (true_peak.rs:173-186,341-352) and the gain it asks for.
// A workgroup owns 64 steps of a group (aligned to absolute multiples of 64); the limiter output of those steps and of the
// 32 before them goes through LDS.
template <int kN>
__device__ __forceinline__ float tp_observe_regs(const float (&h)[kN], int i) {
  // the window of the sample in h[i]: h[i - k] is the sample k steps back
  float peak = fabsf(h[i]);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < kTpTaps; ++k) acc = __builtin_fmaf(AF_TP_FIR[p][k], h[i - k], acc);
    peak = fmaxf(peak, fabsf(acc));
  }
  return peak;
}

__device__ __forceinline__ void stage_f5_body(const StageArgs &a, const float *xin_ring, int bx, int by) {
  __shared__ float xl_t[kTileRows + kTpTaps][kLanes];
  const int g = by;
  const Who w = who(a, g);
  const ChainParams &P = preset(a, g);
  const int wave = threadIdx.x >> 6;
  const int R32 = a.r.rows_f32, R = a.r.rows_f64;
  const int64_t gb32 = (int64_t)g * R32 * kLanes, gb = (int64_t)g * R * kLanes;
  const int la = P.lim.lookahead_samples;
  const double ceil_lin = P.lim.ceiling_linear;
  const float tp_ceiling = P.tp.ceiling_linear;
  const int64_t n0 = a.n0, n_end = a.n0 + a.n;
  const int64_t abs0 = ((n0 >> 6) + bx) * kTileRows;  // first sample of the tile
  // samples abs0 - 32 .. abs0 + 63 of the limiter output (before the stream's first sample the rings hold zeros)
  {  // (all of a wave's 24 row pairs are loaded before the first is used: the launch is short, a load's latency is not)
    constexpr int kMine = (kTileRows + kTpTaps) / 4;
    float delayed[kMine];
    double gain[kMine];
#pragma unroll
    for (int k = 0; k < kMine; ++k) {
      const int64_t n = abs0 - kTpTaps + wave + 4 * k;
      delayed[k] = xin_ring[gb32 + eoff(n - la, w.lane, R32)];
      gain[k] = a.r.g[gb + eoff(n, w.lane, R)];
    }
#pragma unroll
    for (int k = 0; k < kMine; ++k) {
      const int i = wave + 4 * k;
      const int64_t n = abs0 - kTpTaps + i;
      const float o = (float)dclamp((double)delayed[k] * gain[k], -ceil_lin, ceil_lin);
      const float v = finite_f32(o) ? o : 0.0f;  // TruePeakLimiter input scrub, true_peak.rs:342
      xl_t[i][w.lane] = v;
      if (n >= n0 && n < n_end && i >= kTpTaps) a.r.xl[gb32 + eoff(n, w.lane, R32)] = v;
    }
  }
  __syncthreads();
  float h[kTpTaps + 16];  // h[i]: sample abs0 + 16 * wave - 32 + i
#pragma unroll
  for (int i = 0; i < kTpTaps + 16; ++i) h[i] = xl_t[wave * 16 + i][w.lane];
  float itp_v[16], tgt_v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float itp = tp_observe_regs(h, kTpTaps + k);
    float tg = 1.0f;
    if (itp > tp_ceiling) tg = fclamp((tp_ceiling * 0.999f) / itp, 0.0f, 1.0f);
    itp_v[k] = itp;
    tgt_v[k] = tg;
  }
  const int64_t first = abs0 + wave * 16;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t qa = first + 4 * k;
    float *pi = a.r.itp + gb32 + qoff(qa >> 2, R32) + w.lane * kQ, *pt = a.r.tgt + gb32 + qoff(qa >> 2, R32) + w.lane * kQ;
    if (qa >= n0 && qa + 3 < n_end) {
      *reinterpret_cast<float4 *>(pi) = make_float4(itp_v[4 * k], itp_v[4 * k + 1], itp_v[4 * k + 2], itp_v[4 * k + 3]);
      *reinterpret_cast<float4 *>(pt) = make_float4(tgt_v[4 * k], tgt_v[4 * k + 1], tgt_v[4 * k + 2], tgt_v[4 * k + 3]);
    } else {
#pragma unroll
      for (int j = 0; j < kQ; ++j)
        if (qa + j >= n0 && qa + j < n_end) {
          pi[j] = itp_v[4 * k + j];
          pt[j] = tgt_v[4 * k + j];
        }
    }
  }
}

// ============================================================================================ true-peak limiter, serial parts
// (1) the gain (true_peak.rs:353-374) and the limiter's own block figures; (2) the chain output and the block output
// statistics (block_processor.rs:150-170), whose square sum is a recurrence of its own
__device__ __forceinline__ void stage_tp_body(const StageArgs &a, int bx, int /*by*/) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  __builtin_amdgcn_s_setprio(3);
  const int R32 = a.r.rows_f32;
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  Ahead<float> in_itp, in_tgt;
  in_itp.init(a.r.itp, w.g, w.lane, R32, q_first);
  in_tgt.init(a.r.tgt, w.g, w.lane, R32, q_first);
  Out<float> o_g;
  o_g.init(a.r.gt, w.g, w.lane, R32);
  const float rel = P.tp.release_coeff;
  const float one_m_rel = 1.0f - rel;
  float g = a.st32[(int64_t)kTpGain * w.NS + w.sc];
  float tp_in_peak = 0.0f, tp_gmin = 1.0f, tp_limited = 0.0f;
  const int cb = P.control_block;
  BlockStats *stats = a.stats;
  int in_block = 0;
  int64_t b = 0;
  auto block_end = [&]() {
    if (w.valid && stats) {
      BlockStats &row = stats[b * w.NS + w.s];
      row.tp_limiter_input_peak = tp_in_peak;
      row.tp_limiter_gr_db = tp_gmin < 1.0f ? -20.0f * log10f(fmaxf(tp_gmin, 1e-10f)) : 0.0f;
      row.tp_limited_events = tp_limited != 0.0f ? 1u : 0u;
    }
    tp_in_peak = 0.0f;
    tp_gmin = 1.0f;
    tp_limited = 0.0f;
    b += 1;
  };
  for_blocks(q_first, q_last, [&](int64_t qb, auto buf_tag) {
    constexpr int kBuf = decltype(buf_tag)::value;
    const float(&cur_itp)[kU] = in_itp.template buf<kBuf>();
    const float(&cur_tgt)[kU] = in_tgt.template buf<kBuf>();
    auto step = [&](int u) {
      const float itp = cur_itp[u], tg = cur_tgt[u];
      tp_in_peak = fmaxf(tp_in_peak, itp);
      const float released = rel * g + one_m_rel * tg;  // (formed before the comparison picks: same operations, same values)
      const bool limiting = tg < g;
      tp_limited = limiting ? 1.0f : tp_limited;
      g = limiting ? tg : released;
      tp_gmin = fminf(tp_gmin, g);
      o_g.v[u] = g;
    };
    run_block<true>(qb, n0, n, cb, in_block, step, [&] { o_g.store_all(qb); }, [&](int u) { o_g.store_one(qb, u); }, block_end);
    in_itp.template refill<kBuf>(qb);
    in_tgt.template refill<kBuf>(qb);
  });
  if (w.valid) a.st32[(int64_t)kTpGain * w.NS + w.s] = g;
}

template <bool kLim>
__device__ __forceinline__ void stage_out_body(const StageArgs &a, const float *xin_ring, int bx, int /*by*/) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  __builtin_amdgcn_s_setprio(3);
  const int R32 = a.r.rows_f32;
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  Ahead<float> in_x, in_g;
  if (kLim) {
    static_assert(kTpDelay % kQ == 0, "the true-peak delay must be a whole number of quads");
    in_x.init(a.r.xl, w.g, w.lane, R32, q_first, -kTpDelay / kQ);
    in_g.init(a.r.gt, w.g, w.lane, R32, q_first);
  } else {
    in_x.init(xin_ring, w.g, w.lane, R32, q_first);
  }
  Out<float> o_ring;
  o_ring.init(a.r.od, w.g, w.lane, R32);
  const float tp_ceiling = P.tp.ceiling_linear;
  const bool comp_on = (P.flags & kFlagCompressor) != 0;
  double out_sq = 0.0;
  float out_peak = 0.0f, nonfinite = 0.0f;
  const int cb = P.control_block;
  BlockStats *stats = a.stats;
  int in_block = 0;
  int64_t b = 0;
  auto block_end = [&]() {
    if (w.valid && stats) {
      BlockStats &row = stats[b * w.NS + w.s];
      row.output_square_sum = out_sq;
      row.output_sample_peak = out_peak;
      row.non_finite_output = nonfinite != 0.0f ? 1u : 0u;
    }
    out_sq = 0.0;
    out_peak = 0.0f;
    nonfinite = 0.0f;
    b += 1;
  };
  for_blocks(q_first, q_last, [&](int64_t qb, auto buf_tag) {
    constexpr int kBuf = decltype(buf_tag)::value;
    const float(&cur_x)[kU] = in_x.template buf<kBuf>();
    const float(&cur_g)[kU] = in_g.template buf<kBuf>();
    auto step = [&](int u) {
      float o = cur_x[u];
      if (kLim) {
        o = fclamp(o * cur_g[u], -tp_ceiling, tp_ceiling);
        if (!finite_f32(o)) o = 0.0f;
      }
      if (finite_f32(o)) {
        out_sq += (double)o * (double)o;
      } else {
        nonfinite = 1.0f;
      }
      out_peak = fmaxf(out_peak, fabsf(o));
      o_ring.v[u] = o;
    };
    run_block<true>(qb, n0, n, cb, in_block, step, [&] { o_ring.store_all(qb); }, [&](int u) { o_ring.store_one(qb, u); }, block_end);
    in_x.template refill<kBuf>(qb);
    if (kLim) in_g.template refill<kBuf>(qb);
  });
  if (w.valid && !comp_on) a.st64[(int64_t)kCompGr * w.NS + w.s] = 0.0;
}

// ============================================================================================ feed-forward 6
// output-side 4x true peak (TruePeakDetector::process_block, true_peak.rs:205-221; block_processor.rs:159) folded into the
// block maximum, and the chain output back in stream-major order.  Tiles of 64 steps at absolute multiples of 64.
__device__ __forceinline__ void stage_f6_body(const StageArgs &a, float (&tile)[kTileRows][kLanes + 1], int bx, int by) {
  const int g = by;
  const Who w = who(a, g);
  const ChainParams &P = preset(a, g);
  const int wave = threadIdx.x >> 6;
  const int R32 = a.r.rows_f32;
  const float *od = a.r.od + (int64_t)g * R32 * kLanes;
  const int64_t n0 = a.n0, n_end = a.n0 + a.n;
  const int64_t abs0 = ((n0 >> 6) + bx) * kTileRows;
  const int64_t first = abs0 + wave * 16;  // first sample of this wave
  float h[kTpTaps + 16];                   // h[i]: sample first - 32 + i
#pragma unroll
  for (int k = 0; k < (kTpTaps + 16) / kQ; ++k) {
    const Quad<float> x = load_quad(od + qoff(((first - kTpTaps) >> 2) + k, R32) + w.lane * kQ);
#pragma unroll
    for (int j = 0; j < kQ; ++j) h[kQ * k + j] = x.v[j];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) tile[wave * 16 + k][w.lane] = h[kTpTaps + k];
#pragma unroll
  for (int i = 0; i < kTpTaps + 16; ++i)
    if (!finite_f32(h[i])) h[i] = 0.0f;  // the detector scrubs what it is fed (true_peak.rs:212)
  const int cb = P.control_block;
  float m = 0.0f;
  int64_t mb = -1;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t n = first + k;
    if (n >= n0 && n < n_end) {
      const int64_t b = (n - n0) / cb;
      if (b != mb) {
        if (mb >= 0 && w.valid && a.stats)
          atomicMax(reinterpret_cast<unsigned int *>(&a.stats[mb * w.NS + w.s].output_true_peak), __float_as_uint(m));
        m = 0.0f;
        mb = b;
      }
      m = fmaxf(m, tp_observe_regs(h, kTpTaps + k));
    }
  }
  if (mb >= 0 && w.valid && a.stats)
    atomicMax(reinterpret_cast<unsigned int *>(&a.stats[mb * w.NS + w.s].output_true_peak), __float_as_uint(m));
  __syncthreads();
#pragma unroll 4
  for (int r = wave * 16; r < wave * 16 + 16; ++r) {
    const int s = g * kLanes + r;
    const int64_t t = abs0 + w.lane - n0;
    if (s < a.n_streams && t >= 0 && t < a.n) a.out[(int64_t)s * a.stream_stride + t] = tile[w.lane][r];
  }
}


// ============================================================================================ de-esser stages
// The three-band dynamic de-esser (deesser.rs:405-547; af_deesser.hip is the lane-per-stream form: one wave per 64 streams
// walking ~1 200 dependent instructions per sample, 417 ms per 2 s of audio at any batch).  Nothing in it feeds back from the
// audio it produces: detectors (six biquads + four envelopes) -> levels and confidence targets (pure math) -> confidence /
// baseline / raw targets (three small recurrences) -> scaling, reduction smoothing and the 0.001 dB hold (one recurrence) ->
// peaking coefficients (pure math) -> three cascaded dynamic EQs.  So it cuts into stages like the rest of the chain; the
// stages read the same state rows as the lane kernel and use its expressions (af_deesser_math.h): the same bits.
using namespace deess;

// stream-major audio -> the xi ring, scrubbed / clamped (python_api.rs:515-523, routing.rs:802-823); tiles of 64 steps
__device__ __forceinline__ void stage_de0_body(const StageArgs &a, float (&tile)[kTileRows][kLanes + 1], uint32_t flags, int bx, int by) {
  const int g = by;
  const Who w = who(a, g);
  const int wave = threadIdx.x >> 6;
  const int R32 = a.r.rows_f32;
  float *xi = a.r.xi + (int64_t)g * R32 * kLanes;
  const int64_t n0 = a.n0, n_end = a.n0 + a.n;
  const int64_t abs0 = ((n0 >> 6) + bx) * kTileRows;
  const bool scrub = (flags & (kFlagInputScrub | kFlagInputClamp)) != 0, clamp = (flags & kFlagInputClamp) != 0;
#pragma unroll 4
  for (int r = wave * 16; r < wave * 16 + 16; ++r) {  // lane = time: a 256-byte run of one stream
    const int s = g * kLanes + r;
    const int64_t t = abs0 + w.lane - n0;
    float v = 0.0f;
    if (s < a.n_streams && t >= 0 && t < a.n) v = a.in[(int64_t)s * a.stream_stride + t];
    if (scrub && !finite_f32(v)) v = 0.0f;
    if (clamp) v = fclamp(v, -1.0f, 1.0f);
    tile[w.lane][r] = v;
  }
  __syncthreads();
  const int64_t first = abs0 + wave * 16;  // lane = stream: this wave's sixteen steps, a quad at a time
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t qa = first + 4 * k;
    float *p = xi + qoff(qa >> 2, R32) + w.lane * kQ;
    if (qa >= n0 && qa + 3 < n_end) {
      *reinterpret_cast<float4 *>(p) = make_float4(tile[wave * 16 + 4 * k][w.lane], tile[wave * 16 + 4 * k + 1][w.lane],
                                                   tile[wave * 16 + 4 * k + 2][w.lane], tile[wave * 16 + 4 * k + 3][w.lane]);
    } else {
#pragma unroll
      for (int j = 0; j < kQ; ++j)
        if (qa + j >= n0 && qa + j < n_end) p[j] = tile[wave * 16 + 4 * k + j][w.lane];
    }
  }
}

// detector of one band (deesser.rs:405-443): high-pass -> low-pass -> envelope; band 0's wave also keeps the broadband envelope
template <int kBand>
__device__ __forceinline__ void stage_de1_body(const StageArgs &a, const ChainParams &PW, int bx) {
  const Who w = who(a, bx);
  const DeEsserParams &D = PW.deesser;  // the window's own parameter block: the filters' crossfade counters as of its first sample
  __builtin_amdgcn_s_setprio(3);
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  Ahead<float> in;
  in.init(a.r.xi, w.g, w.lane, a.r.rows_f32, q_first);
  Out<double> o_env, o_bb;
  o_env.init(a.r.de_env[kBand], w.g, w.lane, a.r.rows_f64);
  if (kBand == 0) o_bb.init(a.r.de_bb, w.g, w.lane, a.r.rows_f64);
  const double *p = &a.st64[(int64_t)(kDeBand0 + kBand * kDeBandStride) * w.NS + w.sc];
  double env = p[0];
  Bq hp{p[11 * w.NS], p[12 * w.NS], p[13 * w.NS], p[14 * w.NS]}, lp{p[15 * w.NS], p[16 * w.NS], p[17 * w.NS], p[18 * w.NS]};
  double broadband_env = kBand == 0 ? a.st64[(int64_t)kDeBroadbandEnv * w.NS + w.sc] : 0.0;
  const SectionParams sec_hp = D.bands[kBand].detector_hp, sec_lp = D.bands[kBand].detector_lp;  // by value (see stage A)
  const double det_a = D.detector_attack_coeff, det_r = D.detector_release_coeff;
  const bool any_xf = sec_hp.xf_remaining > 0 || sec_lp.xf_remaining > 0;
  const BiquadCoef c_hp = sec_hp.xf_remaining > 0 ? sec_hp.pending : sec_hp.active, c_lp = sec_lp.xf_remaining > 0 ? sec_lp.pending : sec_lp.active;
  int dummy = 0;
  for_blocks(q_first, q_last, [&](int64_t qb, auto buf_tag) {
    constexpr int kBuf = decltype(buf_tag)::value;
    const float(&cur)[kU] = in.template buf<kBuf>();
    auto step = [&](int u) {
      const float input = cur[u];
      if (kBand == 0) {
        broadband_env = smooth_value(broadband_env, (double)fabsf(input), det_a, det_r);
        o_bb.v[u] = broadband_env;
      }
      float side;
      if (any_xf) {  // (wave-uniform: a stream opens with at most a few hundred such samples)
        const int64_t k = qb * kQ + u - n0;  // samples since the window's first
        const int rem_hp = sec_hp.xf_remaining > k ? (int)(sec_hp.xf_remaining - k) : 0;
        const int rem_lp = sec_lp.xf_remaining > k ? (int)(sec_lp.xf_remaining - k) : 0;
        const float sc_hp = section_sample(sec_hp, rem_hp, input, hp);
        side = section_sample(sec_lp, rem_lp, sc_hp, lp);
      } else {
        const float sc_hp = (float)direct(c_hp, (double)input, hp.z1, hp.z2);
        side = (float)direct(c_lp, (double)sc_hp, lp.z1, lp.z2);
      }
      env = smooth_value(env, (double)fabsf(side), det_a, det_r);
      o_env.v[u] = env;
    };
    run_block<false>(
        qb, n0, n, 0, dummy, step,
        [&] {
          o_env.store_all(qb);
          if (kBand == 0) o_bb.store_all(qb);
        },
        [&](int u) {
          o_env.store_one(qb, u);
          if (kBand == 0) o_bb.store_one(qb, u);
        },
        [] {});
    in.template refill<kBuf>(qb);
  });
  if (w.valid) {
    double *q = &a.st64[(int64_t)(kDeBand0 + kBand * kDeBandStride) * w.NS + w.s];
    q[0] = env;
    q[11 * w.NS] = hp.z1; q[12 * w.NS] = hp.z2; q[13 * w.NS] = hp.pz1; q[14 * w.NS] = hp.pz2;
    q[15 * w.NS] = lp.z1; q[16 * w.NS] = lp.z2; q[17 * w.NS] = lp.pz1; q[18 * w.NS] = lp.pz2;
    if (kBand == 0) a.st64[(int64_t)kDeBroadbandEnv * w.NS + w.s] = broadband_env;
  }
}

// levels in dB, voice reference, narrowness, dominance, confidence targets (deesser.rs:425-443,453-470,173-224)
__device__ __forceinline__ void stage_de2_body(const StageArgs &a, int bx, int by) {
  const int g = by;
  const ChainParams &P = preset(a, g);
  const bool auto_enabled = P.deesser.auto_enabled != 0;
  const int R = a.r.rows_f64;
  const int64_t gb = (int64_t)g * R * kLanes;
  const int i = threadIdx.x;
  const int64_t q0 = (a.n0 >> 2) + (int64_t)bx * kFfQuads;
  double e0[kFfQuads], e1[kFfQuads], e2[kFfQuads], bb[kFfQuads];  // (loaded ahead of the arithmetic: see stage F5)
#pragma unroll
  for (int k = 0; k < kFfQuads; ++k) {
    const int64_t row = gb + qoff(q0 + k, R) + i;
    e0[k] = a.r.de_env[0][row];
    e1[k] = a.r.de_env[1][row];
    e2[k] = a.r.de_env[2][row];
    bb[k] = a.r.de_bb[row];
  }
#pragma unroll
  for (int k = 0; k < kFfQuads; ++k) {
    const Elem e = ff_elem(a, q0 + k, i, R);
    if (!e.in) continue;
    const int64_t row = gb + e.idx;
    const double env[3] = {e0[k], e1[k], e2[k]};
    double level_db[3];
    double total_env = 0.0, max_env = 0.0;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      total_env += env[b];
      max_env = fmax(max_env, env[b]);
      level_db[b] = lin2db(env[b], 1e-10);
    }
    const double voice_level = fmax(bb[k] - total_env * kVoiceRefDiscount, 1e-8);
    const double voice_db = lin2db(voice_level, 1e-10);
    const double narrowness = total_env > 1e-10 ? max_env / total_env : 0.0;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const double side_db = level_db[b];
      const double ratio_db = fmax(side_db - voice_db, 0.0);
      const double dominance = max_env > 1e-10 ? sqrt(env[b] / max_env) : 0.0;
      const double ct = confidence_target(side_db, voice_db, narrowness) * dominance;
      a.r.de_ct[b][row] = dclamp(ct, 0.0, 1.0);
      a.r.de_ratio[b][row] = ratio_db;
      a.r.de_aux[b][row] = auto_enabled ? ((voice_db > -55.0 || side_db > -55.0) ? 1.0 : 0.0) : side_db;
    }
  }
}

// one band's confidence, baseline and raw reduction target (deesser.rs:453-517)
template <int kBand>
__device__ __forceinline__ void stage_de3_body(const StageArgs &a, int bx) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  const DeEsserParams &D = P.deesser;
  __builtin_amdgcn_s_setprio(3);
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  const int R = a.r.rows_f64;
  const double *r_ct = a.r.de_ct[kBand] + (int64_t)w.g * R * kLanes, *r_ratio = a.r.de_ratio[kBand] + (int64_t)w.g * R * kLanes;
  const double *r_aux = a.r.de_aux[kBand] + (int64_t)w.g * R * kLanes;
  double *r_tr = a.r.de_tr[kBand] + (int64_t)w.g * R * kLanes;
  double *p = &a.st64[(int64_t)(kDeBand0 + kBand * kDeBandStride) * w.NS + w.sc];
  double confidence = p[w.NS], baseline = p[2 * w.NS];
  // deesser.rs:445-451: the `auto` curve
  const double amount = dclamp(D.auto_amount, 0.0, 1.0);
  const double trigger_offset_db = lerp(8.0, 0.8, amount);
  const double slope = lerp(0.08, 1.9, amount);
  const double auto_cap = lerp(0.8, 14.0, amount);
  const double confidence_floor = lerp(0.28, 0.06, amount);
  const double cap_db = fmin(auto_cap, D.max_reduction_db * 0.75);
  const double det_a = D.detector_attack_coeff, det_r = D.detector_release_coeff;
  const double baseline_fall = D.baseline_fall, baseline_rise = D.baseline_rise, baseline_inactive = D.baseline_inactive;
  const double threshold_db = D.threshold_db, ratio = D.ratio, max_reduction_db = D.max_reduction_db;
  const bool auto_enabled = D.auto_enabled != 0;
  const double cg_start = dclamp(confidence_floor, 0.0, 0.95);
  // quad by quad, the loads two quads ahead (three f64 inputs: whole blocks of sixteen per input would not fit the registers)
  struct Slot {
    Quad<double> ct, ratio, aux;
  };
  auto fetch = [&](int64_t q) {
    const int64_t o = qoff(q, R) + w.lane * kQ;
    return Slot{load_quad(r_ct + o), load_quad(r_ratio + o), load_quad(r_aux + o)};
  };
  const int64_t n_end = n0 + n;
  Slot s0 = fetch(q_first), s1 = fetch(q_first + 1);
  for (int64_t q = q_first; q <= q_last; ++q) {
    const Slot cur = s0;
    s0 = s1;
    s1 = fetch(q + 2);
    double out[kQ];
#pragma unroll
    for (int j = 0; j < kQ; ++j) {
      const int64_t na = q * kQ + j;
      out[j] = 0.0;
      if (na < n0 || na >= n_end) continue;
      const double ratio_db = cur.ratio.v[j];
      confidence = smooth_value(confidence, cur.ct.v[j], det_a, det_r);
      double tr = 0.0;
      if (auto_enabled) {
        const bool voice_active = cur.aux.v[j] != 0.0;
        if (voice_active) {
          const double bt = dclamp(ratio_db * 0.45, 0.0, 24.0);
          const double bc = bt < baseline ? baseline_fall : baseline_rise;
          baseline = bc * baseline + (1.0 - bc) * bt;
        } else {
          baseline *= baseline_inactive;
        }
        const double cg = normalize_range(confidence, cg_start, 1.0);
        const double over = fmax(ratio_db - baseline - trigger_offset_db, 0.0);
        tr = dclamp(over * slope * cg, 0.0, cap_db);
      } else {
        const double side_db = cur.aux.v[j];
        if (side_db > threshold_db) {
          const double ratio_threshold = dclamp((threshold_db + 60.0) * 0.10, 0.0, 6.0);
          const double level_over = side_db - threshold_db;
          const double ratio_over = ratio_db - ratio_threshold;
          if (ratio_over > 0.0) {
            const double over = fmin(level_over, ratio_over);
            const double cg = normalize_range(confidence, 0.22, 1.0);
            tr = dclamp((1.0 - (1.0 / ratio)) * over * cg, 0.0, max_reduction_db * 0.75);
          }
        }
      }
      out[j] = tr;
    }
    double *pd = r_tr + qoff(q, R) + w.lane * kQ;
    if (q * kQ >= n0 && q * kQ + 3 < n_end) {
      *reinterpret_cast<double2 *>(pd) = make_double2(out[0], out[1]);
      *reinterpret_cast<double2 *>(pd + 2) = make_double2(out[2], out[3]);
    } else {
#pragma unroll
      for (int j = 0; j < kQ; ++j)
        if (q * kQ + j >= n0 && q * kQ + j < n_end) pd[j] = out[j];
    }
  }
  if (w.valid) {
    double *q = &a.st64[(int64_t)(kDeBand0 + kBand * kDeBandStride) * w.NS + w.s];
    q[w.NS] = confidence;
    q[2 * w.NS] = baseline;
  }
}

// ---- deesser.rs:518-538 as three kinds of stage (one wave doing all of it was the pipeline's slowest stage by 15 %):
// De4s: the targets scaled to the total budget (in place in the target rings); De4a/b/c: one band's reduction smoothing and
// the 0.001 dB hold on its dynamic EQ's gain; De4t: the total reduction -> the block's figure (feeds no other stage).
__device__ __forceinline__ void stage_de4s_body(const StageArgs &a, int bx) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  __builtin_amdgcn_s_setprio(3);
  const int R = a.r.rows_f64;
  const int64_t n0 = a.n0, n_end = a.n0 + a.n;
  const int64_t q_first = n0 >> 2, q_last = (n_end - 1) >> 2;
  double *r_tr[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) r_tr[b] = a.r.de_tr[b] + (int64_t)w.g * R * kLanes;
  const double max_reduction_db = P.deesser.max_reduction_db;
  struct Slot {
    Quad<double> t[3];
  };
  auto fetch = [&](int64_t q) {
    const int64_t o = qoff(q, R) + w.lane * kQ;
    return Slot{{load_quad(r_tr[0] + o), load_quad(r_tr[1] + o), load_quad(r_tr[2] + o)}};
  };
  Slot s0 = fetch(q_first), s1 = fetch(q_first + 1);
  for (int64_t q = q_first; q <= q_last; ++q) {
    Slot cur = s0;
    s0 = s1;
    s1 = fetch(q + 2);
    bool any = false;
#pragma unroll
    for (int j = 0; j < kQ; ++j) {
      const int64_t na = q * kQ + j;
      if (na < n0 || na >= n_end) continue;
      double target_sum = 0.0;
#pragma unroll
      for (int b = 0; b < 3; ++b) target_sum += cur.t[b].v[j];
      if (target_sum > max_reduction_db && target_sum > 0.0) {
        const double scale = max_reduction_db / target_sum;
#pragma unroll
        for (int b = 0; b < 3; ++b) cur.t[b].v[j] *= scale;
        any = true;
      }
    }
    if (any) {  // (a quad whose targets stayed as they were is not written back)
      const int64_t o = qoff(q, R) + w.lane * kQ;
#pragma unroll
      for (int j = 0; j < kQ; ++j)
        if (q * kQ + j >= n0 && q * kQ + j < n_end) {
#pragma unroll
          for (int b = 0; b < 3; ++b) r_tr[b][o + j] = cur.t[b].v[j];
        }
    }
  }
}

template <int kBand>
__device__ __forceinline__ void stage_de4b_body(const StageArgs &a, int bx) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  const DeEsserParams &D = P.deesser;
  __builtin_amdgcn_s_setprio(3);
  const int64_t n = a.n, n0 = a.n0;
  const int64_t q_first = n0 >> 2, q_last = (n0 + n - 1) >> 2;
  Ahead<double> in;
  in.init(a.r.de_tr[kBand], w.g, w.lane, a.r.rows_f64, q_first);
  Out<double> o_g, o_red;
  o_g.init(a.r.de_gdb[kBand], w.g, w.lane, a.r.rows_f64);
  o_red.init(a.r.de_red[kBand], w.g, w.lane, a.r.rows_f64);
  Out<float> o_upd;
  o_upd.init(a.r.de_upd[kBand], w.g, w.lane, a.r.rows_f32);
  double *p = &a.st64[(int64_t)(kDeBand0 + kBand * kDeBandStride) * w.NS + w.sc];
  double red = p[3 * w.NS], gdb = p[4 * w.NS];
  const double attack = D.attack_coeff, release = D.release_coeff;
  int dummy = 0;
  for_blocks(q_first, q_last, [&](int64_t qb, auto buf_tag) {
    constexpr int kBuf = decltype(buf_tag)::value;
    const double(&cur)[kU] = in.template buf<kBuf>();
    auto step = [&](int u) {
      red = smooth_value(red, cur[u], attack, release);
      const double gain = -red;
      const bool upd = fabs(gdb - gain) > 0.001;  // set_gain_db_immediate (deesser.rs:536-538)
      gdb = upd ? gain : gdb;
      o_red.v[u] = red;
      o_g.v[u] = gdb;
      o_upd.v[u] = upd ? 1.0f : 0.0f;
    };
    run_block<false>(
        qb, n0, n, 0, dummy, step,
        [&] {
          o_red.store_all(qb);
          o_g.store_all(qb);
          o_upd.store_all(qb);
        },
        [&](int u) {
          o_red.store_one(qb, u);
          o_g.store_one(qb, u);
          o_upd.store_one(qb, u);
        },
        [] {});
    in.template refill<kBuf>(qb);
  });
  if (w.valid) {
    double *q = &a.st64[(int64_t)(kDeBand0 + kBand * kDeBandStride) * w.NS + w.s];
    q[3 * w.NS] = red;
    q[4 * w.NS] = gdb;
  }
}

__device__ __forceinline__ void stage_de4t_body(const StageArgs &a, int bx) {
  const Who w = who(a, bx);
  const ChainParams &P = preset(a, w.g);
  const int R = a.r.rows_f64;
  const int64_t n0 = a.n0, n_end = a.n0 + a.n;
  const int64_t q_first = n0 >> 2, q_last = (n_end - 1) >> 2;
  const double *r_red[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) r_red[b] = a.r.de_red[b] + (int64_t)w.g * R * kLanes;
  const double max_reduction_db = P.deesser.max_reduction_db;
  double current_reduction = a.st64[(int64_t)kDeCurrentReduction * w.NS + w.sc];
  const int cb = P.control_block;
  BlockStats *stats = a.stats;
  int in_block = 0;
  int64_t blk = 0;
  struct Slot {
    Quad<double> t[3];
  };
  auto fetch = [&](int64_t q) {
    const int64_t o = qoff(q, R) + w.lane * kQ;
    return Slot{{load_quad(r_red[0] + o), load_quad(r_red[1] + o), load_quad(r_red[2] + o)}};
  };
  Slot s0 = fetch(q_first), s1 = fetch(q_first + 1);
  for (int64_t q = q_first; q <= q_last; ++q) {
    const Slot cur = s0;
    s0 = s1;
    s1 = fetch(q + 2);
#pragma unroll
    for (int j = 0; j < kQ; ++j) {
      const int64_t na = q * kQ + j;
      if (na < n0 || na >= n_end) continue;
      double total_reduction = 0.0;
#pragma unroll
      for (int b = 0; b < 3; ++b) total_reduction += cur.t[b].v[j];
      current_reduction = fmin(total_reduction, max_reduction_db);
      in_block += 1;
      if (in_block == cb || na + 1 == n_end) {  // (wave-uniform) a control block, or the window, ends with this sample
        if (w.valid && stats) stats[blk * w.NS + w.s].deesser_gr_db = (float)current_reduction;
        blk += 1;
        in_block = 0;
      }
    }
  }
  if (w.valid) a.st64[(int64_t)kDeCurrentReduction * w.NS + w.s] = current_reduction;
}

// the peaking coefficients of every gain that changed (Biquad::calculate_coefficients, biquad.rs:109-182)
__device__ __forceinline__ void stage_de5_body(const StageArgs &a, int bx, int by) {
  const int g = by;
  const ChainParams &P = preset(a, g);
  const DeEsserParams &D = P.deesser;
  const int R = a.r.rows_f64, R32 = a.r.rows_f32;
  const int64_t gb = (int64_t)g * R * kLanes, gb32 = (int64_t)g * R32 * kLanes;
  const int i = threadIdx.x;
  const int64_t q0 = (a.n0 >> 2) + (int64_t)bx * kFfQuads;
  float upd[3][kFfQuads];
  double gd[3][kFfQuads];
#pragma unroll
  for (int k = 0; k < kFfQuads; ++k) {
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      upd[b][k] = a.r.de_upd[b][gb32 + qoff(q0 + k, R32) + i];
      gd[b][k] = a.r.de_gdb[b][gb + qoff(q0 + k, R) + i];
    }
  }
#pragma unroll
  for (int k = 0; k < kFfQuads; ++k) {
    const Elem e = ff_elem(a, q0 + k, i, R);
    if (!e.in) continue;
    const int64_t row = gb + e.idx;
#pragma unroll
    for (int b = 0; b < 3; ++b)
      if (upd[b][k] != 0.0f) {
        const BiquadCoef c = peaking(D.bands[b].dyn_cos_omega, D.bands[b].dyn_alpha, gd[b][k]);
        a.r.de_c[b][0][row] = c.b0;
        a.r.de_c[b][1][row] = c.b1;
        a.r.de_c[b][2][row] = c.b2;
        a.r.de_c[b][3][row] = c.a1;
        a.r.de_c[b][4][row] = c.a2;
      }
  }
}

// one dynamic EQ of the cascade (deesser.rs:526-546): band 0 reads the chain input, band i the output of band i - 1.
// Quad by quad with the loads two quads ahead (five coefficient rings: whole blocks of sixteen would not fit the registers).
template <int kBand>
__device__ __forceinline__ void stage_de6_body(const StageArgs &a, const ChainParams &PW, int bx) {
  const Who w = who(a, bx);
  const SectionParams sec = PW.deesser.bands[kBand].dynamic_eq;  // the window's own block (crossfade counters as of its first sample)
  __builtin_amdgcn_s_setprio(3);
  const int R = a.r.rows_f64, R32 = a.r.rows_f32;
  const int64_t n0 = a.n0, n_end = a.n0 + a.n;
  const int64_t q_first = n0 >> 2, q_last = (n_end - 1) >> 2;
  const float *src = (kBand == 0 ? a.r.xi : a.r.de_y[kBand - 1]) + (int64_t)w.g * R32 * kLanes;
  const float *updr = a.r.de_upd[kBand] + (int64_t)w.g * R32 * kLanes;
  float *dst = a.r.de_y[kBand] + (int64_t)w.g * R32 * kLanes;
  const double *cr[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) cr[j] = a.r.de_c[kBand][j] + (int64_t)w.g * R * kLanes;
  double *p = &a.st64[(int64_t)(kDeBand0 + kBand * kDeBandStride) * w.NS + w.sc];
  double cancelled = p[5 * w.NS];
  BiquadCoef dyn{p[6 * w.NS], p[7 * w.NS], p[8 * w.NS], p[9 * w.NS], p[10 * w.NS]};
  Bq eq{p[19 * w.NS], p[20 * w.NS], p[21 * w.NS], p[22 * w.NS]};
  struct Slot {
    Quad<float> x, u;
    Quad<double> c[5];
  };
  auto fetch = [&](int64_t q) {
    Slot s;
    const int64_t o32 = qoff(q, R32) + w.lane * kQ, o64 = qoff(q, R) + w.lane * kQ;
    s.x = load_quad(src + o32);
    s.u = load_quad(updr + o32);
#pragma unroll
    for (int j = 0; j < 5; ++j) s.c[j] = load_quad(cr[j] + o64);
    return s;
  };
  Slot s0 = fetch(q_first), s1 = fetch(q_first + 1);
  for (int64_t q = q_first; q <= q_last; ++q) {
    const Slot cur = s0;
    s0 = s1;
    s1 = fetch(q + 2);
    float y[kQ];
#pragma unroll
    for (int j = 0; j < kQ; ++j) {
      const int64_t na = q * kQ + j;
      y[j] = 0.0f;
      if (na < n0 || na >= n_end) continue;
      if (cur.u.v[j] != 0.0f) {  // set_coefficients_immediate: new coefficients, a pending crossfade is cancelled
        dyn = BiquadCoef{cur.c[0].v[j], cur.c[1].v[j], cur.c[2].v[j], cur.c[3].v[j], cur.c[4].v[j]};
        cancelled = 1.0;
        eq.pz1 = 0.0;
        eq.pz2 = 0.0;
      }
      const int64_t k = na - n0;
      const int rem = sec.xf_remaining > k ? (int)(sec.xf_remaining - k) : 0;
      if (rem > 0 && cancelled == 0.0) {
        y[j] = section_sample(sec, rem, cur.x.v[j], eq);
        if (rem == 1) dyn = sec.pending;
      } else {
        y[j] = (float)direct(dyn, (double)cur.x.v[j], eq.z1, eq.z2);
      }
    }
    float *pd = dst + qoff(q, R32) + w.lane * kQ;
    if (q * kQ >= n0 && q * kQ + 3 < n_end) {
      *reinterpret_cast<float4 *>(pd) = make_float4(y[0], y[1], y[2], y[3]);
    } else {
#pragma unroll
      for (int j = 0; j < kQ; ++j)
        if (q * kQ + j >= n0 && q * kQ + j < n_end) pd[j] = y[j];
    }
  }
  if (w.valid) {
    double *qd = &a.st64[(int64_t)(kDeBand0 + kBand * kDeBandStride) * w.NS + w.s];
    qd[5 * w.NS] = cancelled;
    qd[6 * w.NS] = dyn.b0; qd[7 * w.NS] = dyn.b1; qd[8 * w.NS] = dyn.b2; qd[9 * w.NS] = dyn.a1; qd[10 * w.NS] = dyn.a2;
    qd[19 * w.NS] = eq.z1; qd[20 * w.NS] = eq.z2; qd[21 * w.NS] = eq.pz1; qd[22 * w.NS] = eq.pz2;
  }
}

// ---- all stages of one launch step as roles of TWO dispatches (DiagArgs, af_stages.h): the serial stages (and the
// one-wave-per-block wide one) in workgroups of one wave, the wide stages in workgroups of four.  (One dispatch for both kinds
// costs the wide stages their occupancy -- the serial stages' ~250 registers per lane become every workgroup's: 256 streams
// 36 ms per 10 s instead of 40, but 1024 streams 88 instead of 69 and 4096 streams 330 instead of 225.)
struct RolePick {
  int bx, by;
  int r;
};
__device__ __forceinline__ RolePick pick_role(const DiagArgs &d) {
  const uint32_t b = blockIdx.x;
  int r = 0;
  for (int i = 1; i < d.n_roles; ++i)
    if (b >= d.roles[i].first_block) r = i;  // (uniform)
  const uint32_t local = b - d.roles[r].first_block;
  return RolePick{(int)(local % d.roles[r].gx), (int)(local / d.roles[r].gx), r};
}
__device__ __forceinline__ StageArgs role_args(const DiagArgs &d, const DiagRole &role) {
  StageArgs a = d.base;
  a.n0 = role.win.n0;
  a.n = role.win.n;
  a.stats = role.win.stats;
  a.mk = role.win.mk;
  a.bp = role.win.bp;
  a.vad = role.win.vad;
  a.in = role.win.in;
  a.out = role.win.out;
  return a;
}

__device__ __forceinline__ bool deesser_serial_role(const DiagArgs &d, const DiagRole &role, const StageArgs &a, int bx) {
  // the window's own parameter block (of this group's preset)
  const ChainParams &PW = d.params_eq[role.win.eq_slot + (a.group_preset ? a.group_preset[bx] : 0)];
  switch (role.stage) {
    case kStDe1a: stage_de1_body<0>(a, PW, bx); return true;
    case kStDe1b: stage_de1_body<1>(a, PW, bx); return true;
    case kStDe1c: stage_de1_body<2>(a, PW, bx); return true;
    case kStDe3a: stage_de3_body<0>(a, bx); return true;
    case kStDe3b: stage_de3_body<1>(a, bx); return true;
    case kStDe3c: stage_de3_body<2>(a, bx); return true;
    case kStDe4s: stage_de4s_body(a, bx); return true;
    case kStDe4a: stage_de4b_body<0>(a, bx); return true;
    case kStDe4b: stage_de4b_body<1>(a, bx); return true;
    case kStDe4c: stage_de4b_body<2>(a, bx); return true;
    case kStDe4t: stage_de4t_body(a, bx); return true;
    case kStDe6a: stage_de6_body<0>(a, PW, bx); return true;
    case kStDe6b: stage_de6_body<1>(a, PW, bx); return true;
    case kStDe6c: stage_de6_body<2>(a, PW, bx); return true;
    default: return false;
  }
}

// kDe: the de-esser's serial stages are roles of this dispatch too (a build of its own: the plain chain's keeps its registers)
template <bool kDe>
__global__ __launch_bounds__(64) void stage_diag_serial_kernel(DiagArgs d) {
  const RolePick pk = pick_role(d);
  const DiagRole &role = d.roles[pk.r];
  const StageArgs a = role_args(d, role);
  const int bx = pk.bx, by = pk.by;
  const bool comp = (d.flags & kFlagCompressor) != 0, lim = (d.flags & kFlagLimiter) != 0;
  const float *lim_in = comp ? a.r.xc : a.r.xe;
  if (role.stage == d.debug_skip) return;
  if (kDe && role.stage >= kStDe0) {
    deesser_serial_role(d, role, a, bx);
    return;
  }
  switch (role.stage) {
    case kStEq: {
      EqSystolicArgs ea{d.params_eq + role.win.eq_slot, a.group_preset, a.st64, a.in, nullptr, a.r.xe, d.deesser ? nullptr : a.r.xi, nullptr,
                        a.n, a.stream_stride, a.n0, a.n_streams, a.r.rows_f32, nullptr, d.deesser ? a.r.de_y[2] : nullptr};
      if (role.win.eq_crossfade) eq_systolic_body<false, true>(ea, bx);
      else eq_systolic_body<false, false>(ea, bx);
      break;
    }
    case kStIn: stage_in_body(a, bx, by); break;
    case kStCompA:
      if (d.sidechain) stage_comp_a_body<true>(a, bx, by);
      else stage_comp_a_body<false>(a, bx, by);
      break;
    case kStCompA2:
      if (d.sidechain) stage_comp_a2_body<true>(a, bx, by);
      else stage_comp_a2_body<false>(a, bx, by);
      break;
    case kStCompC: stage_comp_c_body(a, bx, by); break;
    case kStCompE:
      if (d.auto_makeup) {
        if (d.adaptive) stage_comp_e_body<true, true>(a, bx, by);
        else stage_comp_e_body<false, true>(a, bx, by);
      } else {
        if (d.adaptive) stage_comp_e_body<true, false>(a, bx, by);
        else stage_comp_e_body<false, false>(a, bx, by);
      }
      break;
    case kStPow: stage_pow_body(a, bx, by); break;
    case kStMakeup: stage_makeup_body(a, bx, by); break;
    case kStRel: stage_rel_body(a, bx, by); break;
    case kStF4: stage_f4_body(a, lim_in, bx, by); break;
    case kStLim: stage_lim_body(a, bx, by); break;
    case kStTp: stage_tp_body(a, bx, by); break;
    case kStOut:
      if (lim) stage_out_body<true>(a, lim_in, bx, by);
      else stage_out_body<false>(a, lim_in, bx, by);
      break;
    default: break;
  }
}

// the de-esser's serial stages as a kernel of their own (AF_DEESSER_DISPATCH=1: a third dispatch per step; the default runs them
// as roles of the serial dispatch above, side by side with the chain's serial stages)
__global__ __launch_bounds__(64) void stage_diag_deesser_kernel(DiagArgs d) {
  const RolePick pk = pick_role(d);
  const DiagRole &role = d.roles[pk.r];
  const StageArgs a = role_args(d, role);
  deesser_serial_role(d, role, a, pk.bx);
}

__global__ __launch_bounds__(256) void stage_diag_wide_kernel(DiagArgs d) {
  __shared__ float tile[kTileRows][kLanes + 1];  // the transposing stages' (F6, De0) 64-step tile
  const RolePick pk = pick_role(d);
  const DiagRole &role = d.roles[pk.r];
  const StageArgs a = role_args(d, role);
  const int bx = pk.bx, by = pk.by;
  const bool comp = (d.flags & kFlagCompressor) != 0;
  const float *lim_in = comp ? a.r.xc : a.r.xe;
  switch (role.stage) {
    case kStF1: stage_f1_body(a, bx, by); break;
    case kStF2: stage_f2_body(a, bx, by); break;
    case kStFR: stage_fr_body(a, bx, by); break;
    case kStF3: stage_f3_body(a, bx, by); break;
    case kStF3a: stage_f3a_body(a, bx, by); break;
    case kStF5: stage_f5_body(a, lim_in, bx, by); break;
    case kStF6: stage_f6_body(a, tile, bx, by); break;
    case kStDe0: stage_de0_body(a, tile, d.flags, bx, by); break;
    case kStDe2: stage_de2_body(a, bx, by); break;
    case kStDe5: stage_de5_body(a, bx, by); break;
    default: break;
  }
}

}  // namespace


unsigned stage_role_blocks(int stage, int64_t n0, int64_t n, int32_t n_streams, int32_t w_min, unsigned *gy) {
  const unsigned groups = (unsigned)((n_streams + kLanes - 1) / kLanes);
  const unsigned tiles = (unsigned)(((n0 + n - 1) >> 6) - (n0 >> 6) + 1);
  const unsigned quads = (unsigned)(((n0 + n - 1) >> 2) - (n0 >> 2) + 1);
  *gy = groups;
  switch (stage) {
    case kStEq: *gy = 1; return (unsigned)((n_streams + 3) / 4);  // four streams per wave
    case kStF1: case kStF2: case kStF3: case kStF3a: case kStFR: case kStDe2: case kStDe5: return (quads + kFfQuads - 1) / kFfQuads;
    case kStF4: return (unsigned)(n / (w_min > 0 ? w_min : 1) + 2);
    case kStF5: case kStF6: case kStDe0: return tiles;
    default: *gy = 1; return groups;  // serial stages: one workgroup (its first wave) per group
  }
}

int stage_dispatch_kind(int stage) {
  static const bool own_dispatch = [] {  // AF_DEESSER_DISPATCH=1: the de-esser's serial stages as a third dispatch per step
    const char *env = std::getenv("AF_DEESSER_DISPATCH");
    return env && std::atoi(env) != 0;
  }();
  switch (stage) {
    case kStF1: case kStF2: case kStFR: case kStF3: case kStF3a: case kStF5: case kStF6: case kStDe0: case kStDe2: case kStDe5: return 1;
    case kStDe1a: case kStDe1b: case kStDe1c: case kStDe3a: case kStDe3b: case kStDe3c: case kStDe4s: case kStDe4a: case kStDe4b: case kStDe4c: case kStDe4t: case kStDe6a: case kStDe6b: case kStDe6c:
      return own_dispatch ? 2 : 0;
    default: return 0;
  }
}

hipError_t launch_stage_diag(const DiagArgs &d, unsigned total_blocks, int kind, hipStream_t stream) {
  if (total_blocks == 0) return hipSuccess;
  if (kind == 1) hipLaunchKernelGGL(stage_diag_wide_kernel, dim3(total_blocks), dim3(256), 0, stream, d);
  else if (kind == 2) hipLaunchKernelGGL(stage_diag_deesser_kernel, dim3(total_blocks), dim3(64), 0, stream, d);
  else if (d.deesser) hipLaunchKernelGGL(stage_diag_serial_kernel<true>, dim3(total_blocks), dim3(64), 0, stream, d);
  else hipLaunchKernelGGL(stage_diag_serial_kernel<false>, dim3(total_blocks), dim3(64), 0, stream, d);
  return hipGetLastError();
}

}  // namespace af
