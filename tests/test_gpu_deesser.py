"""GPU parity for the de-esser pass (dsp/deesser.rs) and the full golden downstream chain.

The de-esser recomputes peaking-EQ coefficients on the device (exp10 instead of the host's pow(10, x))
and evaluates three log10 per sample, so -- like the compressor -- it is compared within a small
tolerance rather than bit-for-bit: max |err| <= 5e-7, RMS <= 5e-8 (north_star budget: RMS <= 1e-5).
"""
import math
import os

import numpy as np
import pytest

import signals as S

pytestmark = pytest.mark.gpu

MAX_ABS = 5e-7
MAX_RMS = 5e-8


@pytest.fixture(scope="module", params=["quad", "ring-16x4", "lane", "staged"])
def mi(request):
    import mic_eq_mi

    previous = os.environ.get("AF_KERNEL_VARIANT")
    os.environ["AF_KERNEL_VARIANT"] = request.param
    assert mic_eq_mi.CORE_AVAILABLE, "HIP library missing: GPU tests never fall back to the CPU"
    yield mic_eq_mi
    if previous is None:  # the override must not leak into the modules that run after this one
        os.environ.pop("AF_KERNEL_VARIANT", None)
    else:
        os.environ["AF_KERNEL_VARIANT"] = previous


def _err(a, b):
    d = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(d))), float(np.sqrt(np.mean(d * d)))


def _configure_kat(oracle, chain, eng):
    """tests.rs:1786-1808 on both sides."""
    L = oracle.lib()
    for what in ("deesser_enabled", "eq_enabled", "compressor_enabled", "limiter_enabled"):
        chain.set(what, 1)
        getattr(eng, "set_" + what)(1)
    L.afo_deesser_set_auto_enabled(chain.deesser, 1); eng.deesser_set_auto_enabled(1)
    L.afo_deesser_set_auto_amount(chain.deesser, 0.85); eng.deesser_set_auto_amount(0.85)
    L.afo_deesser_set_max_reduction_db(chain.deesser, 10.0); eng.deesser_set_max_reduction_db(10.0)
    for band, (f, g, q) in {2: (180.0, -2.5, 0.8), 6: (2800.0, 3.0, 1.2), 8: (7200.0, 1.5, 1.0)}.items():
        L.afo_eq_set_band_frequency(chain.eq, band, f); L.afo_eq_set_band_gain(chain.eq, band, g); L.afo_eq_set_band_q(chain.eq, band, q)
        eng.eq_set_band_frequency(band, f); eng.eq_set_band_gain(band, g); eng.eq_set_band_q(band, q)
    for name, value in (("threshold", -22.0), ("ratio", 3.5), ("attack_time", 8.0), ("release_time", 160.0), ("makeup_gain", 8.0)):
        getattr(L, f"afo_compressor_set_{name}")(chain.compressor, value)
        getattr(eng, f"compressor_set_{name}")(value)
    L.afo_compressor_set_adaptive_release(chain.compressor, 1); eng.compressor_set_adaptive_release(1)
    L.afo_limiter_set_ceiling(chain.limiter, -6.0); L.afo_limiter_set_release_time(chain.limiter, 55.0)
    eng.limiter_set_ceiling(-6.0); eng.limiter_set_release_time(55.0)


def test_golden_downstream_chain_kat_on_gpu(mi, oracle):
    """The reference's golden KAT (tests.rs:1784-1885), all four stages, through the HIP path:
    its pinned aggregates, and sample parity with the oracle that reproduces them."""
    chain = oracle.Chain(48_000.0)
    eng = mi.Engine(48_000.0, 2)
    _configure_kat(oracle, chain, eng)
    eng.set_control_block_samples(480)
    x = S.kat_signal(300)
    want = x.copy()
    rows = [chain.process_block(want[b * 480 : (b + 1) * 480]) for b in range(300)]
    batch = np.stack([x, x * np.float32(0.7)])
    out_a = eng.process(batch[:, : 37 * 480])
    st_a = eng.block_stats()
    out_b = eng.process(batch[:, 37 * 480 :])
    st_b = eng.block_stats()
    y = np.concatenate([out_a, out_b], axis=1)[0]
    stats = np.concatenate([st_a, st_b], axis=0)
    max_abs, rms_err = _err(y, want)
    assert max_abs <= MAX_ABS and rms_err <= MAX_RMS, (max_abs, rms_err)

    # the reference's own assertions, at the reference's tolerances
    y64 = y.astype(np.float64)
    rms = math.sqrt(float(np.mean(y64 * y64)))
    weighted = math.fsum(y64 * ((np.arange(y.size) % 997) + 1).astype(np.float64))
    assert abs(rms - 0.185_715_270_552) <= 1.0e-6
    assert abs(float(np.abs(y).max()) - 0.500_814_14) <= 2.0e-6
    assert abs(weighted - (-4_246.481_547_342)) <= 0.05
    col = stats[:, 0]
    assert abs(float(col["compressor_gain_reduction_db"].max()) - 8.687_991) <= 0.001
    assert abs(float(col["deesser_gain_reduction_db"].max()) - 10.0) <= 0.001
    max_lim = max(float(col["limiter_peak_gain_reduction_db"].max()), float(col["true_peak_limiter_gain_reduction_db"].max()))
    assert abs(max_lim - 4.348_602) <= 0.001
    assert 20 <= int(col["true_peak_limited_events"].sum()) <= 24
    for idx, exp in zip((1_000, 10_000, 50_000, 100_000), [-0.038_492_45, 0.185_469_2, 0.200_082_9, -0.093_881_376]):
        assert abs(float(y[idx]) - exp) <= 2.0e-5
    # per-block de-esser metering against the oracle
    dees = np.array([r.deesser_gain_reduction_db for r in rows], dtype=np.float32)
    assert np.max(np.abs(col["deesser_gain_reduction_db"] - dees)) <= 1e-4
    insq = np.array([float(np.sum(x[b * 480 : (b + 1) * 480].astype(np.float64) ** 2)) for b in range(300)])
    assert np.allclose(col["input_square_sum"], insq, rtol=1e-12, atol=0.0)
    eng.close()


@pytest.mark.parametrize("eq_first", [False, True])
@pytest.mark.parametrize("auto", [True, False])
def test_deesser_orders_and_manual_mode(mi, oracle, eq_first, auto):
    """Both stage orders (routing.rs eq_before_deesser), auto and threshold/ratio modes, moved detector
    band (a pending coefficient crossfade on the nine de-esser filters), ragged batch and block sizes."""
    if eq_first and os.environ["AF_KERNEL_VARIANT"] in ("lane", "quad"):
        pytest.skip("EQ-before-de-esser needs the ring kernel's pre-pass")
    if os.environ["AF_KERNEL_VARIANT"] == "staged":
        pytest.skip("this test runs the realtime front end without the suppressor (kernels 1-3); the stage pipeline's de-esser "
                    "stages have test_deesser_stages_equal_the_lane_kernel below")
    L = oracle.lib()
    n_streams, n = 67, 48_000 + 333
    audio = np.stack([S.kat_signal(101, *S.stream_params(s))[:n] for s in range(n_streams)])
    eng = mi.Engine(48_000.0, n_streams)
    for what in ("deesser_enabled", "eq_enabled", "compressor_enabled", "limiter_enabled"):
        getattr(eng, "set_" + what)(1)
    eng.set_eq_before_deesser(int(eq_first))
    eng.set_input_clamp_enabled(1); eng.set_prefilter_enabled(1, 1)

    def configure_chain(chain):
        for what in ("deesser_enabled", "eq_enabled", "compressor_enabled", "limiter_enabled"):
            chain.set(what, 1)
        chain.set("eq_before_deesser", int(eq_first))
        d = chain.deesser
        L.afo_deesser_set_auto_enabled(d, int(auto)); L.afo_deesser_set_auto_amount(d, 0.7)
        L.afo_deesser_set_low_cut_hz(d, 3500.0); L.afo_deesser_set_high_cut_hz(d, 9000.0)
        L.afo_deesser_set_threshold_db(d, -40.0); L.afo_deesser_set_ratio(d, 6.0)
        L.afo_deesser_set_attack_ms(d, 1.0); L.afo_deesser_set_release_ms(d, 60.0)
        L.afo_deesser_set_max_reduction_db(d, 8.0)
        L.afo_eq_set_band_gain(chain.eq, 7, 4.0)
        L.afo_compressor_set_threshold(chain.compressor, -24.0)

    eng.deesser_set_auto_enabled(int(auto)); eng.deesser_set_auto_amount(0.7)
    eng.deesser_set_low_cut_hz(3500.0); eng.deesser_set_high_cut_hz(9000.0)
    eng.deesser_set_threshold_db(-40.0); eng.deesser_set_ratio(6.0)
    eng.deesser_set_attack_ms(1.0); eng.deesser_set_release_ms(60.0)
    eng.deesser_set_max_reduction_db(8.0)
    eng.eq_set_band_gain(7, 4.0)
    eng.compressor_set_threshold(-24.0)
    eng.set_control_block_samples(441)
    out = np.concatenate([eng.process(audio[:, :10_000]), eng.process(audio[:, 10_000:])], axis=1)
    worst = (0.0, 0.0)
    any_reduction = 0.0
    for s in (0, 1, 31, 63, 64, 66):
        chain = oracle.Chain(48_000.0)
        configure_chain(chain)
        want = oracle.prefilter(np.clip(audio[s], -1.0, 1.0))
        pos = 0
        while pos < n:
            st = chain.process_block(want[pos : pos + 441])
            any_reduction = max(any_reduction, st.deesser_gain_reduction_db)
            pos += 441
        e = _err(out[s], want)
        worst = (max(worst[0], e[0]), max(worst[1], e[1]))
    assert worst[0] <= MAX_ABS and worst[1] <= MAX_RMS, worst
    assert any_reduction > 0.5  # the de-esser really acted
    eng.close()


@pytest.mark.parametrize("auto", [True, False])
def test_deesser_stages_equal_the_lane_kernel(oracle, auto):
    """The de-esser as stages of the stage pipeline (af_stages.hip: loader | three detectors | levels and confidence targets |
    three confidence / baseline recurrences | scaling + reduction smoothing + gain hold | coefficients | three cascaded dynamic
    EQs) against the lane-per-stream pass: the same expressions on the same state rows, so audio and block rows agree BIT FOR
    BIT -- auto and threshold / ratio modes, moved detector band (a pending coefficient crossfade on the nine de-esser filters),
    ragged batch, ragged control block, a call that ends inside a block; AUTO routes the configuration to the stages."""
    import mic_eq_mi as mi

    n_streams, n = 67, 24_000 + 333
    audio = np.stack([S.kat_signal(51, *S.stream_params(s))[:n] for s in range(n_streams)])

    def run(kernel):
        previous = os.environ.pop("AF_KERNEL_VARIANT", None)
        try:
            eng = mi.Engine(48_000.0, n_streams)
        finally:
            if previous is not None:
                os.environ["AF_KERNEL_VARIANT"] = previous
        for what in ("deesser_enabled", "eq_enabled", "compressor_enabled", "limiter_enabled"):
            getattr(eng, "set_" + what)(1)
        eng.set_input_clamp_enabled(1)
        eng.deesser_set_auto_enabled(int(auto)); eng.deesser_set_auto_amount(0.7)
        eng.deesser_set_low_cut_hz(3500.0); eng.deesser_set_high_cut_hz(9000.0)
        eng.deesser_set_threshold_db(-40.0); eng.deesser_set_ratio(6.0)
        eng.deesser_set_attack_ms(1.0); eng.deesser_set_release_ms(60.0)
        eng.deesser_set_max_reduction_db(8.0)
        eng.eq_set_band_gain(7, 4.0)
        eng.compressor_set_threshold(-24.0)
        eng.set_control_block_samples(441)
        eng.set_kernel(kernel)
        outs, rows, used = [], [], []
        for lo, hi in ((0, 10_000), (10_000, n)):
            outs.append(eng.process(audio[:, lo:hi]))
            rows.append(eng.block_stats().copy())
            used.append(eng.last_kernel())
        eng.close()
        return np.concatenate(outs, axis=1), np.concatenate(rows, axis=0), used

    lane = run(1)
    auto_routed = run(0)
    assert set(lane[2]) == {1} and set(auto_routed[2]) == {4}, (lane[2], auto_routed[2])
    assert np.array_equal(lane[0].view(np.uint32), auto_routed[0].view(np.uint32)), float(np.abs(lane[0] - auto_routed[0]).max())
    for name in lane[1].dtype.names:
        assert lane[1][name].tobytes() == auto_routed[1][name].tobytes(), name
    assert float(lane[1]["deesser_gain_reduction_db"].max()) > 0.5  # the de-esser really acted
    # ... and against the oracle, as every kernel
    chain = oracle.Chain(48_000.0)
    L = oracle.lib()
    for what in ("deesser_enabled", "eq_enabled", "compressor_enabled", "limiter_enabled"):
        chain.set(what, 1)
    d = chain.deesser
    L.afo_deesser_set_auto_enabled(d, int(auto)); L.afo_deesser_set_auto_amount(d, 0.7)
    L.afo_deesser_set_low_cut_hz(d, 3500.0); L.afo_deesser_set_high_cut_hz(d, 9000.0)
    L.afo_deesser_set_threshold_db(d, -40.0); L.afo_deesser_set_ratio(d, 6.0)
    L.afo_deesser_set_attack_ms(d, 1.0); L.afo_deesser_set_release_ms(d, 60.0)
    L.afo_deesser_set_max_reduction_db(d, 8.0)
    L.afo_eq_set_band_gain(chain.eq, 7, 4.0)
    L.afo_compressor_set_threshold(chain.compressor, -24.0)
    want = np.clip(audio[66], -1.0, 1.0).copy()
    pos = 0
    while pos < n:
        chain.process_block(want[pos : pos + 441])
        pos += 441
    e = _err(auto_routed[0][66], want)
    assert e[0] <= MAX_ABS and e[1] <= MAX_RMS, e
