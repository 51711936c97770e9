// af_stages.h -- arguments of the stage-pipeline form of the dynamics chain (af_stages.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "af_device.h"

namespace af {

// Hand-over buffers between the stage kernels.  Every one is a ring over ABSOLUTE sample index per 64-stream group, `rows`
// samples long (a power of two), time-major in quads: element (group g, sample n, lane l) sits at
// g * rows * 64 + ((n >> 2) mod (rows / 4)) * 256 + l * 4 + (n & 3), i.e. a lane's four consecutive samples are one
// 16- or 32-byte run and a wave whose lane is the stream moves four steps with one vector load or store.  History a
// stage needs from before its window (limiter lookahead, the 32-tap true-peak windows, the 20-sample delay) is simply
// older rows of the same ring, also across calls; a fresh engine starts from zeroed rings.
struct StageRings {
  float *xi;                                     // chain input after scrub / clamp (what the block input statistics see)
  float *xe;                                     // compressor input (after the EQ)
  double *d, *pr;                                // side-chain signal, its presence-weighted form
  double *low_e, *voiced_e, *pres_e, *rms_e;     // the four envelopes, per sample
  double *ipk_db, *rms_db, *w_db;                // instantaneous peak / rms level, detector weight (dB)
  double *peak_db;                               // log-domain peak envelope
  double *target;                                // static gain-reduction target (dB)
  double *gr;                                    // smoothed gain reduction (dB)
  double *glin;                                  // auto-makeup only: its linear gain, 10^(-gr / 20)
  double *fast_r, *slow_r, *tgt_ms;              // adaptive release only: the envelopes each step found, the release time they ask for
  float *xc;                                     // limiter input (compressor output)
  float *sfx;                                    // suffix maxima of |xc| inside lookahead-aligned blocks
  double *tg;                                    // limiter target gain
  double *g;                                     // limiter gain
  float *xl;                                     // true-peak limiter input (limiter output)
  float *itp, *tgt;                              // 4x true peak of xl and the target gain it asks for
  float *gt;                                     // true-peak limiter gain
  float *od;                                     // chain output (time-major), input of the output-side detector
  // de-esser stages (allocated only when the de-esser is on)
  double *de_env[3], *de_bb;                     // band envelopes, broadband envelope
  double *de_ct[3], *de_ratio[3], *de_aux[3];    // confidence target, band-to-voice ratio (dB), voice-active flag / band level (dB)
  double *de_tr[3];                              // raw reduction targets (dB)
  double *de_red[3];                             // smoothed reductions (dB)
  double *de_gdb[3];                             // the dynamic EQs' gains after the 0.001 dB hold
  float *de_upd[3];                              // 1.0: the band's coefficients change at this sample
  double *de_c[3][5];                            // the coefficients they change to
  float *de_y[3];                                // audio after dynamic EQ 0, 1, 2
  int32_t rows_f32, rows_f64;                    // ring lengths
};

struct StageArgs {
  const ChainParams *params;    // device; an array when `group_preset` is set
  const int32_t *group_preset;  // [groups] or null
  double *st64;
  float *st32;
  BlockStats *stats;            // rows of this window: [block][stream]
  double *mk;                   // [blocks of this window][stream]: linear makeup gain in force during the block
  double *bp;                   // auto-makeup: [blocks of this window][stream] square sum of the compressor's input
  const double *vad;            // auto-makeup: [blocks of this window][stream] speech posteriors, or null
  const float *in;              // stream-major audio at the two ends of the pipeline (`in`: read by the EQ stage only)
  float *out;
  int64_t stream_stride;
  int64_t n;                    // samples per stream in this window
  int64_t n0;                   // absolute index of the window's first sample
  int32_t n_streams;
  int32_t w_min;                // smallest lookahead + 1 over the presets (sizes the limiter stage's grid)
  StageRings r;
};

enum StageId : int {
  kStEq = 0,   // scrub / clamp + 10-band EQ (af_eq_systolic.hip): stream-major audio -> the xi and xe rings
  kStIn,       // serial: block input statistics
  kStCompA,    // serial: side-chain high-pass, low-band envelope, presence signal
  kStCompA2,   // serial: voiced / presence / rms envelopes
  kStF1,       // levels in dB, detector weight
  kStCompC,    // serial: log-domain peak envelope
  kStF2,       // blended detector level -> gain-reduction target
  kStCompE,    // serial: release meter + gain-reduction smoothing, makeup gain per block
  kStF3,       // apply gain
  kStPow,      // serial, auto-makeup: block power of the compressor's input (what the two-launch form's pre-pass measures)
  kStF3a,      // auto-makeup: linear gain of the gain reduction
  kStMakeup,   // serial, auto-makeup: makeup gain, K-weighted loudness meter, the controller (compressor.rs:528-653,700-774)
  kStFR,       // adaptive release: target release time from the envelopes
  kStRel,      // serial, adaptive release: release-time smoothing (feeds no other stage)
  kStF4,       // limiter: sliding maximum over the lookahead window -> target gain
  kStLim,      // serial: limiter gain
  kStF5,       // limiter output, input-side 4x true peak, true-peak target gain
  kStTp,       // serial: true-peak gain
  kStOut,      // serial: chain output, block output statistics
  kStF6,       // output-side 4x true peak, time-major -> stream-major
  // the de-esser (deesser.rs:405-547) ahead of the EQ: every recurrence a stage of its own
  kStDe0,      // stream-major audio -> the xi ring (scrub / clamp)
  kStDe1a, kStDe1b, kStDe1c,  // serial, one per band: detector high-pass -> low-pass -> envelope (band 0 also the broadband envelope)
  kStDe2,      // levels in dB, voice reference, narrowness, dominance, confidence targets
  kStDe3a, kStDe3b, kStDe3c,  // serial, one per band: confidence, baseline, raw reduction target
  kStDe4s,     // serial: the raw targets scaled to the total budget (in place)
  kStDe4a, kStDe4b, kStDe4c,  // serial, one per band: reduction smoothing, the 0.001 dB hold on the dynamic EQ's gain
  kStDe4t,     // serial: total reduction -> the block's figure (feeds no other stage)
  kStDe5,      // the peaking coefficients of every changed gain
  kStDe6a, kStDe6b, kStDe6c,  // serial, cascaded: the three dynamic EQs
  kStCount
};

// ---- one launch for a whole diagonal of the (stage, window) grid ---------------------------------------------------------
// Launch step j runs stage k on window j - skew(k) for every stage at once (roles of a dispatch, picked by block index; two
// dispatches per step: one-wave workgroups for the serial stages, four-wave ones for the wide stages); a stage's inputs were
// written by earlier steps, so stream order is all the synchronisation there is.
struct DiagWin {               // what differs from window to window
  int64_t n0, n;
  BlockStats *stats;
  double *mk, *bp;
  const double *vad;
  const float *in;
  float *out;
  int32_t eq_slot;             // index into params_eq (0)
  int32_t eq_crossfade;        // a coefficient crossfade is pending in that block
};
struct DiagRole {
  int32_t stage;               // StageId
  uint32_t first_block, gx;    // blocks [first_block, next role's first_block): block b -> (bx, by) = ((b - first) % gx, (b - first) / gx)
  DiagWin win;
};
struct DiagArgs {
  StageArgs base;              // everything that is the same for all windows (rings, state planes, parameter block)
  const ChainParams *params_eq;  // the parameter block the EQ stage reads (it moves from window to window while a crossfade runs)
  DiagRole roles[kStCount];
  int32_t n_roles;
  uint32_t flags;              // chain flags the pipeline was planned for
  int32_t sidechain, adaptive, auto_makeup;  // compressor switches (they pick code paths)
  int32_t deesser;             // the de-esser stages run ahead of the EQ (which then reads their output ring)
  int32_t debug_skip;          // timing probes only (AF_STAGE_SKIP=<StageId>): that stage returns at once; results are garbage
};
// `kind` (stage_dispatch_kind): 0 = the one-wave roles (serial stages, F4, the EQ), 1 = the wide stages (workgroups of four
// waves), 2 = the de-esser's serial stages (one wave each, a kernel of their own)
int stage_dispatch_kind(int stage);
hipError_t launch_stage_diag(const DiagArgs &d, unsigned total_blocks, int kind, hipStream_t stream);
// blocks a role needs for a window: `gx` (the launch uses gx * gy blocks, gy = groups except for the EQ stage)
unsigned stage_role_blocks(int stage, int64_t n0, int64_t n, int32_t n_streams, int32_t w_min, unsigned *gy);

}  // namespace af
