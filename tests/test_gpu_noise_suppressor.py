"""The reference's own wrapper tests for the RNNoise suppressor (a3/a4), through the C ABI on the GPU:
rust-core/src/dsp/rnnoise.rs:330-451, rust-core/src/audio/processor/tests.rs:1887-1928.  The counts and transfer-function
pins are exact; audio is compared with the CPU restatement (core parity vs nnnoiseless stays UNPINNED, see
tests/test_oracle_rnnoise_wrapper.py).
"""
import ctypes as C

import numpy as np
import pytest

import signals as S
from test_oracle_rnnoise_wrapper import check_soft_clip_transfer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mi():
    import mic_eq_mi

    assert mic_eq_mi.CORE_AVAILABLE
    return mic_eq_mi


def _device_scale(mi, values):
    from mic_eq_mi import _lib

    x = np.ascontiguousarray(values, dtype=np.float32)
    y = np.zeros_like(x)
    fp = C.POINTER(C.c_float)
    _lib.check(_lib.load().af_suppressor_debug_scale_for_model(x.ctypes.data_as(fp), y.ctypes.data_as(fp), x.size, 0))
    return y


def test_soft_clip_transfer_pins_on_device(mi, oracle):
    """test_rnnoise_model_input_soft_clip_transfer (rnnoise.rs:335-352) on the device function the pre-pass kernel uses,
    then the device against the oracle bit for bit over a sweep that crosses the knee, the limit and the specials."""
    check_soft_clip_transfer(lambda v: float(_device_scale(mi, [v])[0]))
    sweep = np.concatenate([np.linspace(-3.0, 3.0, 20001), np.linspace(0.97, 1.01, 4001), [0.0, -0.0, 1e-30, 50.0, -50.0]]).astype(np.float32)
    got = _device_scale(mi, sweep)
    want = np.array([oracle.scale_sample_for_model(float(v)) for v in sweep], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    specials = _device_scale(mi, [np.nan, np.inf, -np.inf])
    assert np.array_equal(specials, np.zeros(3, dtype=np.float32))


def test_frame_buffering_counts(mi):
    """test_rnnoise_frame_buffering, rnnoise.rs:355-370."""
    p = mi.new_noise_suppression_engine("rnnoise", 3)
    p.push_samples(np.zeros((3, 400), dtype=np.float32))
    p.process_frames()
    assert p.available_samples() == 0
    assert p.pending_input() == 400
    p.push_samples(np.zeros((3, 100), dtype=np.float32))
    p.process_frames()
    assert p.available_samples() == 480
    assert p.pending_input() == 20
    assert p.model_type() == mi.NoiseModel.RNNOISE and p.latency_samples() == 480
    assert p.backend_available() and not p.backend_failed() and p.backend_error() is None
    p.close()


def test_bypass_and_hot_samples(mi):
    """test_rnnoise_bypass + test_disabled_rnnoise_preserves_hot_input_samples, rnnoise.rs:372-399."""
    p = mi.NoiseSuppressor("rnnoise", 2)
    p.set_enabled(False)
    assert not p.is_enabled()
    p.push_samples(np.ones((2, 100), dtype=np.float32))
    p.process_frames()
    assert p.available_samples() == 100
    p.close()
    p = mi.NoiseSuppressor("rnnoise", 2)
    p.set_enabled(False)
    hot = np.array([[1.25, -1.5, 0.25, -0.75], [-9.0, 3.5, np.float32(1e-40), 2.0]], dtype=np.float32)
    p.push_samples(hot)
    p.process_frames()
    out = p.pop_samples(4)
    assert out.shape == (2, 4) and np.array_equal(out.view(np.uint32), hot.view(np.uint32))
    p.close()


def test_strength_getter_setter_and_mix(mi, oracle):
    """test_rnnoise_strength_getter_setter + test_rnnoise_wet_dry_mix, rnnoise.rs:401-434."""
    p = mi.NoiseSuppressor("rnnoise", 1)
    assert p.get_strength() == 1.0
    p.set_strength(0.5)
    assert p.get_strength() == 0.5
    p.set_strength(1.5)
    assert p.get_strength() == 1.0
    p.set_strength(-0.5)
    assert p.get_strength() == 0.0
    p.close()
    p = mi.NoiseSuppressor("rnnoise", 1, weight_seed=0x5EED)
    p.set_strength(0.5)
    p.push_samples(np.full((1, 480), 0.5, dtype=np.float32))
    p.process_frames()
    assert p.available_samples() == 480
    got = p.pop_samples(480)[0]
    ref = oracle.RNNoiseProcessor(0.5)
    ref.push_samples(np.full(480, 0.5, dtype=np.float32))
    ref.process_frames()
    want = ref.read_samples(480)
    assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 1e-5
    p.close()


def test_clipped_input_stays_finite(mi):
    """test_rnnoise_output_stays_finite_for_clipped_input, rnnoise.rs:436-450."""
    p = mi.NoiseSuppressor("rnnoise", 1)
    for n in range(480):
        p.push_samples(np.array([[1.0 if n % 2 == 0 else -1.0]], dtype=np.float32))
    p.process_frames()
    out = p.pop_samples(480)
    assert out.shape == (1, 480)
    assert np.all(np.isfinite(out)) and float(np.abs(out).max()) <= 2.0
    p.close()


def test_full_rt_block_without_short_write(mi):
    """tests.rs:1887-1928 (RT_PROCESS_BUFFER_CAPACITY = 8192)."""
    block = np.zeros((2, 8192), dtype=np.float32)
    p = mi.new_noise_suppression_engine("rnnoise", 2)
    assert p.push_samples(block) == 8192
    p.process_frames()
    expected = (8192 // 480) * 480
    assert p.available_samples() == expected
    assert p.pending_input() == 8192 - expected
    assert p.pop_samples(expected).shape == (2, expected)
    p.close()
    p = mi.new_noise_suppression_engine("rnnoise", 2)
    p.set_enabled(False)
    assert p.push_samples(block) == 8192
    p.process_frames()
    assert p.available_samples() == 8192 and p.pending_input() == 0
    p.close()


def test_trait_surface_matches_restatement_over_ragged_pushes(mi, oracle):
    """Ragged pushes (the realtime loop reads "whatever the ring holds", dsp_loop.rs:960), a live strength change, a bypass
    interval and a soft reset: counts equal the oracle's RNNoiseProcessor exactly at every step, audio to 1e-5."""
    x = S.batch_signal(3, 30)  # 14 400 samples per stream
    p = mi.NoiseSuppressor("rnnoise", 3, weight_seed=0x5EED)
    refs = [oracle.RNNoiseProcessor(1.0, 0x5EED) for _ in range(3)]
    pos = 0
    step = 0
    for size in (400, 100, 1000, 920, 37, 4800, 3, 2000, 480, 960, 1500):
        if step == 4:
            p.set_strength(0.4)
            for r in refs:
                r.set_strength(0.4)
        if step == 6:
            p.set_enabled(False)
            for r in refs:
                r.set_enabled(False)
        if step == 7:
            p.set_enabled(True)
            for r in refs:
                r.set_enabled(True)
        if step == 9:
            p.soft_reset()
            for r in refs:
                r.soft_reset()
        chunk = x[:, pos : pos + size]
        pos += size
        assert p.push_samples(chunk) == refs[0].push_samples(chunk[0])
        for k in (1, 2):
            refs[k].push_samples(chunk[k])
        p.process_frames()
        for r in refs:
            r.process_frames()
        assert p.available_samples() == refs[0].available_samples(), step
        assert p.pending_input() == refs[0].pending_input(), step
        take = p.available_samples() if step % 2 else min(p.available_samples(), 700)
        got = p.pop_samples(take)
        for k in range(3):
            want = refs[k].read_samples(take)
            assert got[k].size == want.size
            if want.size:
                assert float(np.max(np.abs(got[k].astype(np.float64) - want))) <= 1e-5, (step, k)
        step += 1
    drained = p.drain_pending_input()
    assert drained.shape[1] == refs[0].pending_input()
    assert np.array_equal(drained[1], refs[1].drain_pending_input())
    assert p.pending_input() == 0
    p.close()


def test_engine_keeps_the_remainder_between_calls(mi, oracle):
    """The engine-level form of the same ring (the path the chain hangs off): 400 -> 0 out, 400 pending; +100 -> 480 out,
    20 pending; and two calls of 1000 + 920 samples equal one call of 1920, bit for bit, with the chain behind it."""
    from mic_eq_mi import mic_eq_core as core

    eng = core.Engine(48_000.0, 5)
    eng.set_eq_enabled(0)
    eng.set_limiter_enabled(0)
    eng.set_compressor_enabled(0)
    eng.set_suppressor_enabled(1)
    eng.set_control_block_samples(480)
    out = eng.process(np.zeros((5, 400), dtype=np.float32))
    assert out.shape == (5, 0) and eng.pending_input() == 400 and eng.last_output_samples() == 0
    out = eng.process(np.zeros((5, 100), dtype=np.float32))
    assert out.shape == (5, 480) and eng.pending_input() == 20 and eng.last_output_samples() == 480
    eng.close()

    x = S.batch_signal(5, 4)  # 1920 samples
    settings = S.limiter_settings(2.0)

    def engine():
        e = core.Engine(48_000.0, 5)
        core.configure_auto_eq_chain(e, 48_000.0, S.LIMITER_BANDS, settings)
        e.set_prefilter_enabled(1, 1)
        e.set_suppressor_enabled(1)
        e.set_control_block_samples(480)
        return e

    e1 = engine()
    whole = e1.process(x)
    e1.close()
    e2 = engine()
    a = e2.process(x[:, :1000])
    assert a.shape == (5, 960) and e2.pending_input() == 40
    b = e2.process(x[:, 1000:])
    assert b.shape == (5, 960) and e2.pending_input() == 0
    e2.close()
    got = np.concatenate([a, b], axis=1)
    assert np.array_equal(got.view(np.uint32), whole.view(np.uint32))
    want = oracle.simulate_auto_eq_chain(oracle.suppressor_process(oracle.prefilter(x[2]), 1.0), 48_000, S.LIMITER_BANDS,
                                         dict(settings))["output_audio"]
    # (the oracle's simulator cuts 960-sample blocks; with the compressor's per-block bookkeeping off the audio is the same)
    d = whole[2].astype(np.float64) - np.asarray(want, dtype=np.float64)
    assert float(np.sqrt(np.mean(d * d))) <= 1e-5


def test_ragged_call_keeps_the_window_pipeline(mi, oracle):
    """A call whose frame count is not a whole number of control blocks (111 frames, blocks of 960 samples) runs as
    aligned windows plus one short final window (af_api.cpp) and must equal frame-by-frame processing; the last block
    row is the short block."""
    from mic_eq_mi import mic_eq_core as core

    x = S.batch_signal(4, 111)
    settings = S.limiter_settings(2.0)
    eng = core.Engine(48_000.0, 4)
    core.configure_auto_eq_chain(eng, 48_000.0, S.LIMITER_BANDS, settings)  # control block 960
    eng.set_suppressor_enabled(1)
    got = eng.process(x)
    rows = eng.block_stats()
    eng.close()
    assert got.shape == x.shape and rows.shape == (56, 4)
    for s in (0, 3):
        sup = oracle.suppressor_process(x[s], 1.0)
        want = oracle.simulate_auto_eq_chain(sup, 48_000, S.LIMITER_BANDS, dict(settings))
        d = got[s].astype(np.float64) - np.asarray(want["output_audio"], dtype=np.float64)
        assert float(np.sqrt(np.mean(d * d))) <= 1e-5
    assert np.allclose(rows["output_square_sum"].sum(axis=0), (got.astype(np.float64) ** 2).sum(axis=1), rtol=1e-9)


def test_deesser_suppressor_and_prefilter_together(mi, oracle):
    """De-esser (default order: ahead of the EQ) + RNNoise suppressor + DC block / 80 Hz high-pass, two calls of several
    windows each.  The front-end memories belong to the suppressor's pre-pass, which runs windows ahead of the de-esser
    pass: a de-esser launch that wrote its (stale) copy back would corrupt them and the next window's DC block would
    restart from old state (af_api.cpp / af_deesser.hip)."""
    from mic_eq_mi import mic_eq_core as core

    x = (S.batch_signal(6, 240) + np.float32(0.03)).astype(np.float32)  # a DC offset makes the DC block's state matter
    settings = dict(S.limiter_settings(2.0), deesser_enabled=True, deesser_auto_amount=0.85, deesser_max_reduction_db=10.0)
    eng = core.Engine(48_000.0, 6)
    core.configure_auto_eq_chain(eng, 48_000.0, S.LIMITER_BANDS, settings)
    eng.set_prefilter_enabled(1, 1)
    eng.set_suppressor_enabled(1)
    a = eng.process(x[:, : 130 * 480])
    b = eng.process(x[:, 130 * 480 :])
    eng.close()
    got = np.concatenate([a, b], axis=1)
    worst = 0.0
    for s in (0, 5):
        sup = oracle.suppressor_process(oracle.prefilter(x[s]), 1.0)
        want = oracle.simulate_auto_eq_chain(sup, 48_000, S.LIMITER_BANDS, dict(settings))
        assert want["deesser_gain_reduction_db"] > 1.0  # the de-esser is really working on this signal
        d = got[s].astype(np.float64) - np.asarray(want["output_audio"], dtype=np.float64)
        worst = max(worst, float(np.sqrt(np.mean(d * d))))
    assert worst <= 1e-5, worst


def test_gpu_against_the_scalar_order_oracle_with_pitch_decisions(mi, oracle):
    """GPU vs the implementation-independent restatement (published scalar C summation order, afo_rnn_eval_order = 1)
    on streams at four input levels; the number of frames whose pitch index differs is counted and bounded, not hidden
    behind a worst-sample tolerance."""
    from mic_eq_mi import mic_eq_core as core

    gains = (1.0, 2.0, 0.2, 0.02)
    x = np.stack([(S.kat_signal(250, *S.stream_params(3 * k)) * np.float32(g)).astype(np.float32) for k, g in enumerate(gains)])
    eng = core.Engine(48_000.0, len(gains))
    eng.set_eq_enabled(0)
    eng.set_compressor_enabled(0)
    eng.set_limiter_enabled(0)
    eng.set_suppressor_enabled(1)
    eng.set_control_block_samples(480)
    eng.suppressor_set_trace_enabled(1)
    got = eng.process(x)
    trace = eng.suppressor_trace()
    eng.close()
    assert trace.shape == (250, len(gains), 2)
    flips = 0
    for k in range(len(gains)):
        want, pitch, silence = oracle.suppressor_process_traced(x[k], 1.0, 0x5EED, eval_order=1)
        d = got[k].astype(np.float64) - want.astype(np.float64)
        assert float(np.sqrt(np.mean(d * d))) <= 1e-5, k
        assert np.array_equal(trace[:, k, 0], silence), k
        flips += int(np.count_nonzero(trace[:, k, 1] != pitch))
    print(f"pitch decisions differing GPU vs scalar-order oracle: {flips} of {250 * len(gains)} frames")
    assert flips <= 10


def test_a_refused_call_leaves_the_frame_ring_untouched(mi):
    """Everything that can refuse a call is checked before the engine's frame ring or a stream is touched (af_api.cpp,
    af_engine_process_device): a call refused for its stride, or for a VAD array of the wrong length, leaves
    af_engine_pending_input as it was, and the stream continues as if the call had not been made."""
    import torch

    from mic_eq_mi import _lib
    from mic_eq_mi import mic_eq_core as core

    x = torch.from_numpy(S.batch_signal(3, 6)).cuda()  # 2880 samples
    settings = dict(S.limiter_settings(2.0), compressor_auto_makeup_enabled=True)

    def engine():
        e = core.Engine(48_000.0, 3)
        core.configure_auto_eq_chain(e, 48_000.0, S.LIMITER_BANDS, settings)
        e.set_suppressor_enabled(1)
        e.set_control_block_samples(480)
        return e

    def run(e, lo, hi):
        # (one stride serves both buffers: rows long enough for the frames this call may complete, pending ones included)
        stride = (hi - lo) + 480
        seg = torch.zeros((3, stride), dtype=torch.float32, device="cuda")
        seg[:, : hi - lo] = x[:, lo:hi]
        out = torch.zeros((3, stride), dtype=torch.float32, device="cuda")
        e.process_device(seg.data_ptr(), out.data_ptr(), hi - lo, stride, 0, 0)
        torch.cuda.synchronize()
        n = e.last_output_samples()
        return out[:, :n].cpu().numpy()

    ref = engine()
    want = np.concatenate([run(ref, 0, 400), run(ref, 400, 1500), run(ref, 1500, 2880)], axis=1)
    ref.close()

    e = engine()
    a = run(e, 0, 400)
    assert a.shape[1] == 0 and e.pending_input() == 400
    # (1) this call would complete 1440 samples per stream but says its rows are only 1100 long
    with pytest.raises(Exception) as err:
        e.process_device(x.data_ptr(), x.data_ptr(), 1100, 1100, 0, 0)
    assert "stream_stride" in str(err.value)
    assert e.pending_input() == 400
    # (2) auto-makeup evidence at the wrong cadence: 3 blocks will complete, 5 probabilities are offered
    vad = np.full(5, 0.5)
    _lib.check(e._lib.af_compressor_set_activity_evidence(e._h, vad.ctypes.data_as(C.POINTER(C.c_double)), 5, 0, 0.8, -60.0, 0.5))
    with pytest.raises(Exception) as err:
        run(e, 400, 1500)
    assert "VAD" in str(err.value)
    assert e.pending_input() == 400
    _lib.check(e._lib.af_compressor_set_activity_evidence(e._h, None, 0, 0, 0.0, 1.0, 0.0))  # evidence off again
    b = run(e, 400, 1500)
    c = run(e, 1500, 2880)
    e.close()
    got = np.concatenate([a, b, c], axis=1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
