"""GPU RNNoise suppressor vs the CPU restatement (oracle/af_rnnoise.c).

PARITY UNPINNED against the reference's `nnnoiseless 0.5.2` (crate and trained weights are not in
the checkout): these tests pin the GPU kernels to this repository's restatement of the published
RNNoise algorithm on seeded synthetic weights.  Tolerance: the two sides run different FFT
factorizations in f32 (mixed radix on the CPU, 15x8x8 on the GPU), so spectra agree to ~1e-6
relative; a pitch decision can flip on a near-tie, which shows up as one frame of larger error.
We therefore require per-sample RMS error <= 1e-5 (north_star budget) on +-1 full scale and bound
the worst sample at 1e-5 (measured: 2.4e-7).
"""
import numpy as np
import pytest

import signals as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mi():
    import mic_eq_mi

    assert mic_eq_mi.CORE_AVAILABLE
    return mic_eq_mi


WORST_SAMPLE = 1e-5  # measured: 1.2e-7 .. 2.4e-7 (round 1 allowed 2e-3 "in case a pitch decision flips": none does, and the full-size test counts them)


def _check(got, want):
    assert got.shape == want.shape
    assert np.all(np.isfinite(got))
    d = got.astype(np.float64) - want.astype(np.float64)
    rms = float(np.sqrt(np.mean(d * d)))
    worst = float(np.max(np.abs(d)))
    print(f"suppressor GPU vs restatement: rms {rms:.3e}, worst sample {worst:.3e}")
    assert rms <= 1e-5, (rms, worst)
    assert worst <= WORST_SAMPLE, (rms, worst)
    return rms, worst


def test_benchmark_protocol_matches_restatement(mi, oracle):
    x = S.kat_signal(300)  # 3 s: voiced tones + sibilant bursts + noise
    want = oracle.rnnoise_benchmark_frames(x, 0x5EED)
    got = mi.suppress(x, 1.0, 0x5EED, raw_protocol=True)
    _check(got, want)
    # one frame of latency (rnnoise.rs:313-315): the first frame is the analysis window's fade-in only
    assert float(np.abs(want[:480]).max()) < float(np.abs(want[480:960]).max()) + 1.0


def test_wrapper_soft_clip_and_wet_dry_mix(mi, oracle):
    x = (S.kat_signal(120, *S.stream_params(3)) * np.float32(2.6)).astype(np.float32)  # drives the soft clip
    for strength in (1.0, 0.35):
        want = oracle.suppressor_process(x, strength, 77)
        got = mi.suppress(x, strength, 77)
        _check(got, want)


def test_batch_of_streams_and_state_across_calls(mi, oracle):
    audio = S.batch_signal(20, 130)  # 20 streams (2 network tiles: 16 + 4), 1.3 s
    want = np.stack([oracle.suppressor_process(audio[s], 1.0, 0x5EED) for s in range(20)])
    eng = mi.Engine(48_000.0, 20)
    eng.set_eq_enabled(0)
    eng.set_limiter_enabled(0)
    eng.set_suppressor_enabled(1)
    eng.set_control_block_samples(480)
    a = eng.process(audio[:, : 70 * 480])   # two calls, several windows inside each
    b = eng.process(audio[:, 70 * 480 :])
    eng.close()
    got = np.concatenate([a, b], axis=1)
    _check(got, want)


def test_suppressor_then_chain(mi, oracle):
    """Full north_star order: suppressor -> EQ -> compressor -> limiter -> true-peak limiter."""
    x = S.kat_signal(200)
    settings = S.limiter_settings(2.0)
    sup = oracle.suppressor_process(x, 1.0, 0x5EED)
    want = oracle.simulate_auto_eq_chain(sup, 48_000, S.LIMITER_BANDS, dict(settings, compressor_adaptive_release=False))
    from mic_eq_mi import mic_eq_core as core

    eng = core.Engine(48_000.0, 1)
    core.configure_auto_eq_chain(eng, 48_000.0, S.LIMITER_BANDS, settings)
    eng.set_suppressor_enabled(1)
    got = eng.process(x.reshape(1, -1))[0]
    eng.close()
    d = got.astype(np.float64) - want["output_audio"].astype(np.float64)
    assert float(np.sqrt(np.mean(d * d))) <= 2e-5


def test_front_end_then_suppressor(mi, oracle):
    """Realtime order (dsp_loop.rs:1222-1250,1521-1599): clamp -> DC block + 80 Hz HP -> suppressor, strength < 1 so the
    dry path (the front end's output) is audible in the mix."""
    x = (S.kat_signal(100, *S.stream_params(9)) * np.float32(1.7) + np.float32(0.05)).astype(np.float32)
    import ctypes as C

    L = oracle.lib()
    L.afo_sanitize_and_clamp.argtypes = [C.POINTER(C.c_float), C.c_size_t]
    y = x.copy()
    L.afo_sanitize_and_clamp(y.ctypes.data_as(C.POINTER(C.c_float)), y.size)
    y = oracle.prefilter(y)
    want = oracle.suppressor_process(y, 0.6, 0x5EED)
    eng = mi.Engine(48_000.0, 1)
    eng.set_eq_enabled(0)
    eng.set_limiter_enabled(0)
    eng.set_input_clamp_enabled(1)
    eng.set_prefilter_enabled(1, 1)
    eng.set_suppressor_enabled(1)
    eng.set_suppressor_strength(0.6)
    eng.set_control_block_samples(480)
    got = eng.process(x.reshape(1, -1))[0]
    eng.close()
    _check(got, want)


def test_a_ragged_call_returns_whole_frames(mi):
    """rnnoise.rs:114-164: 500 samples in -> one frame out, 20 samples wait (tests/test_gpu_noise_suppressor.py holds the
    reference's own counts)."""
    eng = mi.Engine(48_000.0, 1)
    eng.set_suppressor_enabled(1)
    out = eng.process(np.zeros((1, 500), dtype=np.float32))
    assert out.shape == (1, 480) and eng.pending_input() == 20
    eng.close()


def test_ramped_window_schedule_matches_restatement(mi, oracle):
    """A call long enough for the ramped window schedule (windows of 4, 8, 16, 20 ..., 16, 8, 4 frames,
    af_api.cpp) must give what frame-by-frame processing gives: the windows are an execution detail."""
    audio = S.batch_signal(18, 260)  # 2.6 s: 4+8+16 | 20 x 10, 4 | 16+8+4
    want = np.stack([oracle.suppressor_process(audio[s], 1.0, 0x5EED) for s in range(audio.shape[0])])
    got = mi.suppress(audio, 1.0, 0x5EED)
    _check(got, want)


@pytest.mark.parametrize("kernel,adaptive", [(0, False), (2, False), (2, True)], ids=["auto", "token-ring", "token-ring-adaptive"])
def test_north_star_chain_with_auto_makeup_behind_the_suppressor(mi, oracle, kernel, adaptive):
    """north_star's chain as literally named: DC block / 80 Hz high-pass -> RNNoise suppressor -> 10-band EQ -> compressor WITH
    auto-makeup (compressor.rs:598-653,700-722: per-block activity -> momentary loudness -> makeup) -> lookahead limiter ->
    true-peak limiter, in the realtime stage order (dsp_loop.rs:1222-1250).  70 streams (two chain workgroups, the second
    ragged), two calls of several suppressor windows each, against oracle.suppressor_process o simulate_auto_eq_chain.
    kernel 0 = AUTO (the stage pipeline at this batch), 2 = the token-ring kernel (what batch 4096 runs): its windows take
    the systolic EQ kernel as their pre-pass (block powers) and ONE chain launch each."""
    from mic_eq_mi import mic_eq_core as core

    n_streams, frames_a, frames_b = 70, 130, 110
    x = (S.batch_signal(n_streams, frames_a + frames_b) + np.float32(0.02)).astype(np.float32)
    settings = dict(S.limiter_settings(2.0), compressor_auto_makeup_enabled=True, compressor_target_lufs=-16.0,
                    compressor_adaptive_release=adaptive)
    bands = list(S.LIMITER_BANDS)
    bands[3] = (bands[3][0], 4.0, 1.2)
    bands[7] = (bands[7][0], -3.0, 0.9)
    eng = core.Engine(48_000.0, n_streams)
    core.configure_auto_eq_chain(eng, 48_000.0, bands, settings)
    eng.set_prefilter_enabled(1, 1)
    eng.set_suppressor_enabled(1)
    eng.set_kernel(kernel)
    a = eng.process(x[:, : frames_a * 480])
    rows_a = eng.block_stats()
    used = eng.last_kernel()
    b = eng.process(x[:, frames_a * 480 :])
    rows_b = eng.block_stats()
    eng.close()
    assert used == (4 if kernel == 0 else 2)
    got = np.concatenate([a, b], axis=1)
    makeup = np.concatenate([rows_a["compressor_makeup_gain_db"], rows_b["compressor_makeup_gain_db"]], axis=0)  # [block][stream]
    worst = 0.0
    for s in (0, 1, 63, 64, 69):
        sup = oracle.suppressor_process(oracle.prefilter(x[s]), 1.0)
        want = oracle.simulate_auto_eq_chain(sup, 48_000, bands, dict(settings))
        d = got[s].astype(np.float64) - np.asarray(want["output_audio"], dtype=np.float64)
        rms = float(np.sqrt(np.mean(d * d)))
        worst = max(worst, rms)
        assert rms <= 1e-5, (s, rms)
        assert float(makeup[:, s].max()) > 1.0, s  # the controller really moved
    print(f"north_star chain with auto-makeup (kernel {kernel}, adaptive {adaptive}): worst RMS vs oracle {worst:.3e}")
