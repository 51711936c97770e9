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

#include <type_traits>

#include "af_dsp.h"
#include "af_eq_systolic_body.h"

namespace af {

template <bool kStats, bool kXf, bool kPower = false>
__global__ __launch_bounds__(64) void eq_systolic_kernel(EqSystolicArgs a) {
  eq_systolic_body<kStats, kXf, kPower>(a, blockIdx.x);
}

// `audio`: stream-major output (may be `in`); or null and `ring` / `ring_in` / `ring_rows` / `n0`: the stage pipeline's rings.
// `stats` null: no block input statistics.  `crossfade`: some section has a coefficient crossfade pending.
// `block_power` ([block][stream], with `stats`): also the square sum of every control block of the filtered samples -- the
// launch is then the pre-pass of an auto-makeup window.
hipError_t launch_eq_systolic(const ChainParams *d_params, const int32_t *d_group_preset, double *st64, const float *in, float *audio,
                              float *ring, float *ring_in, int32_t ring_rows, int64_t n0, BlockStats *stats, bool crossfade,
                              int64_t n_samples, int64_t stream_stride, int32_t n_streams, hipStream_t stream, double *block_power) {
  EqSystolicArgs a{d_params, d_group_preset, st64, in, audio, ring, ring_in, stats, n_samples, stream_stride, n0, n_streams, ring_rows,
                   block_power};
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
