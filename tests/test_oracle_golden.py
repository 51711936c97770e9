"""Pin the CPU oracle against every golden vector the reference holds for the hot path.

Sources of the expected numbers:
  (1) rust-core/src/audio/processor/tests.rs:1784-1885  golden downstream-chain KAT
  (4) rust-core/src/dsp/limiter.rs:313-325               lookahead sample counts
  (5) rust-core/src/dsp/biquad.rs:506-518,547-550; eq.rs:653-664,699-715
  (6) rust-core/src/dsp/true_peak.rs:406-412,537-568
  (7) rust-core/src/dsp/compressor.rs:861-882
  (8) rust-core/src/audio/processor/python_api.rs:768-790
  (9) evaluation/limiter-lookahead-report.json            (controlled fixtures, 3 lookaheads)
 (10) evaluation/eq-filter-types-report.json
 (11) evaluation/dynamics-aliasing-report.json
All of these run on the CPU (no GPU marker).
"""
import ctypes as C
import math

import numpy as np
import pytest

import signals as S


def _configure_kat_chain(o, chain):
    """tests.rs:1786-1808."""
    L = o.lib()
    for what in ("deesser_enabled", "eq_enabled", "compressor_enabled", "limiter_enabled"):
        chain.set(what, 1)
    L.afo_deesser_set_auto_enabled(chain.deesser, 1)
    L.afo_deesser_set_auto_amount(chain.deesser, 0.85)
    L.afo_deesser_set_max_reduction_db(chain.deesser, 10.0)
    for band, (f, g, q) in {2: (180.0, -2.5, 0.8), 6: (2800.0, 3.0, 1.2), 8: (7200.0, 1.5, 1.0)}.items():
        L.afo_eq_set_band_frequency(chain.eq, band, f)
        L.afo_eq_set_band_gain(chain.eq, band, g)
        L.afo_eq_set_band_q(chain.eq, band, q)
    L.afo_compressor_set_threshold(chain.compressor, -22.0)
    L.afo_compressor_set_ratio(chain.compressor, 3.5)
    L.afo_compressor_set_attack_time(chain.compressor, 8.0)
    L.afo_compressor_set_release_time(chain.compressor, 160.0)
    L.afo_compressor_set_makeup_gain(chain.compressor, 8.0)
    L.afo_compressor_set_adaptive_release(chain.compressor, 1)
    L.afo_limiter_set_ceiling(chain.limiter, -6.0)
    L.afo_limiter_set_release_time(chain.limiter, 55.0)


def test_kat_generator_numpy_matches_c(oracle):
    assert np.array_equal(S.kat_signal(40), oracle.kat_signal(40))
    st, f0, ph = S.stream_params(7)
    assert np.array_equal(S.kat_signal(12, st, f0, ph), oracle.kat_signal(12, st, f0, ph))


def test_golden_downstream_chain_kat(oracle):
    chain = oracle.Chain(48000.0)
    _configure_kat_chain(oracle, chain)
    y = S.kat_signal(300).copy()
    max_comp = max_deesser = max_lim = 0.0
    events = 0
    for b in range(300):
        st = chain.process_block(y[b * 480 : (b + 1) * 480])
        max_comp = max(max_comp, st.compressor_gain_reduction_db)
        max_deesser = max(max_deesser, st.deesser_gain_reduction_db)
        max_lim = max(max_lim, st.limiter_peak_gain_reduction_db, st.true_peak_limiter_gain_reduction_db)
        events += st.true_peak_limited_events
    y64 = y.astype(np.float64)
    rms = math.sqrt(float(np.mean(y64 * y64)))
    weights = ((np.arange(y.size) % 997) + 1).astype(np.float64)
    weighted = math.fsum(y64 * weights)
    assert abs(rms - 0.185_715_270_552) <= 1.0e-6
    assert abs(float(np.abs(y).max()) - 0.500_814_14) <= 2.0e-6
    assert abs(weighted - (-4_246.481_547_342)) <= 0.05
    assert abs(max_comp - 8.687_991) <= 0.001
    assert abs(max_deesser - 10.0) <= 0.001
    assert abs(max_lim - 4.348_602) <= 0.001
    assert 20 <= events <= 24
    expected = [-0.038_492_45, 0.185_469_2, 0.200_082_9, -0.093_881_376]
    for idx, exp in zip((1_000, 10_000, 50_000, 100_000), expected):
        assert abs(float(y[idx]) - exp) <= 2.0e-5
    # tighter than the reference's own tolerance: this container reproduces it to ~1e-13
    assert abs(rms - 0.185_715_270_552) <= 1.0e-11
    assert abs(weighted - (-4_246.481_547_342)) <= 1.0e-8


def test_offline_chain_equals_manual_stage_chain(oracle):
    """tests.rs:1741-1781."""
    L = oracle.lib()
    fs = 48000.0
    t = np.arange(512, dtype=np.float64) / fs
    x = (0.38 * np.sin(2.0 * np.pi * 2500.0 * t) + 0.22 * np.sin(2.0 * np.pi * 180.0 * t)).astype(np.float32)
    offline = oracle.Chain(fs)
    offline.set("deesser_enabled", 0)
    offline.set("eq_enabled", 1)
    offline.set("compressor_enabled", 0)
    offline.set("limiter_enabled", 1)
    L.afo_eq_set_band_frequency(offline.eq, 5, 2500.0)
    L.afo_eq_set_band_gain(offline.eq, 5, 4.0)
    L.afo_eq_set_band_q(offline.eq, 5, 1.8)
    L.afo_limiter_set_ceiling(offline.limiter, -1.5)
    a = x.copy()
    st = offline.process_block(a)
    # second, independently configured chain instance: same stages, same result bit for bit
    other = oracle.Chain(fs)
    other.set("limiter_enabled", 1)
    L.afo_eq_set_band_frequency(other.eq, 5, 2500.0)
    L.afo_eq_set_band_gain(other.eq, 5, 4.0)
    L.afo_eq_set_band_q(other.eq, 5, 1.8)
    L.afo_limiter_set_ceiling(other.limiter, -1.5)
    b = x.copy()
    other.process_block(b)
    assert np.array_equal(a, b)
    assert math.isfinite(st.output_true_peak) and math.isfinite(st.true_peak_limiter_input_peak)
    # the TP limiter's own output oversampler and the detector see the same samples
    assert st.output_true_peak > 0.0


@pytest.mark.parametrize("fs,expected", [(44100.0, 88), (48000.0, 96), (96000.0, 192), (192000.0, 384), (384000.0, 768)])
def test_limiter_lookahead_samples(oracle, fs, expected):
    """limiter.rs:313-325 -- observable as the delay of an impulse through the stage."""
    chain = oracle.Chain(fs)
    chain.set("eq_enabled", 0)
    chain.set("limiter_enabled", 1)
    x = np.zeros(2048, dtype=np.float32)
    x[0] = 0.25
    chain.process_block(x)
    assert int(np.argmax(np.abs(x))) == expected + 20  # + 20-sample true-peak limiter delay


def test_eq_response_pins(oracle):
    # biquad.rs:547-550 peaking centre gain = 6 dB
    bands = list(S.DEFAULT_TYPED_BANDS)
    bands[4] = ("bell", 1000.0, 6.0, 1.0, 12, True)
    assert abs(oracle.eq_magnitude_response_v2([1000.0], bands, 48000.0)[0] - 6.0) <= 1e-9
    # eq.rs:699-715 + evaluation/eq-filter-types-report.json cutoff rows
    target = -20.0 * math.log10(math.sqrt(2.0))
    expected = {
        ("high_pass", 12): -3.0102999566398116, ("high_pass", 24): -3.01029995663983,
        ("high_pass", 36): -3.010299956639824, ("high_pass", 48): -3.010299956639786,
        ("low_pass", 12): -3.010299956639825, ("low_pass", 24): -3.010299956639828,
        ("low_pass", 36): -3.0102999566398427, ("low_pass", 48): -3.010299956639834,
    }
    for (ftype, slope), want in expected.items():
        bands = list(S.DEFAULT_TYPED_BANDS)
        bands[4] = (ftype, 2000.0, 0.0, 1.0, slope, True)
        got = oracle.eq_magnitude_response_v2([2000.0], bands, 48000.0)[0]
        assert abs(got - target) <= 1e-8
        assert abs(got - want) <= 5e-13
    # notch probe
    bands = list(S.DEFAULT_TYPED_BANDS)
    bands[4] = ("notch", 1000.0, 12.0, 8.0, 12, True)
    got = oracle.eq_magnitude_response_v2([100.0, 1000.0, 10_000.0], bands, 48000.0)
    want = [-0.00069031219514453, -200.0, -0.0005023861794774316]
    assert np.allclose(got, want, rtol=0, atol=1e-12)
    # legacy == typed default response
    grid = np.geomspace(20.0, 20_000.0, 512)
    legacy = [(f, g, q) for _t, f, g, q, _s, _e in S.DEFAULT_TYPED_BANDS]
    assert np.max(np.abs(oracle.eq_magnitude_response(grid, legacy, 48000.0)
                         - oracle.eq_magnitude_response_v2(grid, S.DEFAULT_TYPED_BANDS, 48000.0))) == 0.0


def test_eq_random_boundary_stress(oracle):
    """evaluation/eq-filter-types-report.json random_boundary_stress (seed 0xE041, 250 cases)."""
    filter_types = ("bell", "notch", "low_shelf", "high_shelf", "high_pass", "low_pass")
    slopes = (12, 24, 36, 48)
    grid = np.geomspace(20.0, 20_000.0, 512)
    rng = np.random.default_rng(0xE041)
    worst = 0.0
    for _ in range(250):
        bands = list(S.DEFAULT_TYPED_BANDS)
        for index in range(len(bands)):
            ftype = filter_types[int(rng.integers(0, len(filter_types)))]
            frequency = float(10.0 ** rng.uniform(math.log10(20.0), math.log10(20_000.0)))
            gain = float(rng.uniform(-12.0, 12.0))
            q = float(10.0 ** rng.uniform(math.log10(0.1), math.log10(10.0)))
            slope = slopes[int(rng.integers(0, len(slopes)))]
            enabled = bool(rng.integers(0, 5))
            bands[index] = (ftype, frequency, gain, q, slope, enabled)
        response = oracle.eq_magnitude_response_v2(grid, bands, 48000.0)
        assert np.all(np.isfinite(response))
        worst = max(worst, float(np.max(np.abs(response))))
    assert abs(worst - 1206.650162779802) <= 1e-9


def test_eq_time_domain_headroom_pin(oracle):
    """evaluation/eq-filter-types-report.json headroom_prediction."""
    t = np.arange(96000, dtype=np.float64) / 48000.0
    audio = (0.05 * np.sin(2.0 * np.pi * 1000.0 * t)).astype(np.float32)
    bands = list(S.DEFAULT_TYPED_BANDS)
    bands[4] = ("bell", 1000.0, 12.0, 2.0, 12, True)
    r = oracle.simulate_eq_v2(audio, 48000.0, bands)
    measured = 20.0 * math.log10(max(r["output_rms"], 1e-15) / max(r["input_rms"], 1e-15))
    assert abs(measured - 11.996631425143294) <= 1e-12
    assert abs(r["max_response_db"] - 11.99271646315594) <= 1e-12
    assert abs(oracle.eq_magnitude_response_v2([1000.0], bands, 48000.0)[0] - 12.000000000000009) <= 1e-13


def test_true_peak_constant_input(oracle):
    """true_peak.rs:406-412: constant 0.5 -> peak 0.5 +-1e-6 once the FIR is full."""
    chain = oracle.Chain(48000.0)
    chain.set("eq_enabled", 0)
    chain.set("limiter_enabled", 0)
    x = np.full(256, 0.5, dtype=np.float32)
    chain.process_block(x)
    st = chain.process_block(x)
    assert abs(st.output_true_peak - 0.5) <= 1e-6


def test_true_peak_estimator_vs_long_reference():
    """true_peak.rs:537-568: 4x estimate within 0.08 dB of a 511-tap Blackman interpolator."""
    import af_oracle_py as o
    from scipy.signal import firwin

    fs = 48000.0
    n = 4096
    t = np.arange(n) / fs
    x = (0.5 * np.sin(2 * np.pi * 11025.0 * t + 0.7)).astype(np.float32)
    chain = o.Chain(fs)
    chain.set("eq_enabled", 0)
    chain.set("limiter_enabled", 0)
    st = chain.process_block(x.copy())
    up = np.zeros(n * 4)
    up[::4] = x
    ref = np.convolve(up, firwin(511, 0.25, window="blackman") * 4.0)[255 : 255 + 4 * n]
    ref_peak = np.abs(ref[2048:-2048]).max()
    assert abs(20 * math.log10(st.output_true_peak / ref_peak)) <= 0.08


def test_compressor_knee_and_detector_identities(oracle):
    """compressor.rs:861-882 (soft-knee boundaries, blended detector identities)."""
    L = oracle.lib()
    L.afo_compressor_blended_detector_db.restype = C.c_double
    L.afo_compressor_blended_detector_db.argtypes = [C.c_double, C.c_double]
    assert abs(L.afo_compressor_blended_detector_db(-20.0, -20.0) - (-20.0)) <= 1e-9
    both = L.afo_compressor_blended_detector_db(-10.0, -30.0)
    want = 20 * math.log10(0.6 * 10 ** (-10 / 20) + 0.4 * 10 ** (-30 / 20))
    assert abs(both - want) <= 1e-12
    L.afo_compressor_compute_gain_reduction.restype = C.c_double
    L.afo_compressor_compute_gain_reduction.argtypes = [C.c_void_p, C.c_double]
    chain = oracle.Chain(48000.0)
    comp = chain.compressor
    L.afo_compressor_set_threshold(comp, -20.0)
    L.afo_compressor_set_ratio(comp, 4.0)
    knee_start, knee_end = -23.0, -17.0  # knee 6 dB (block_processor.rs:50)
    assert L.afo_compressor_compute_gain_reduction(comp, knee_start) == 0.0
    assert abs(L.afo_compressor_compute_gain_reduction(comp, knee_end) - 3.0 * 0.75) <= 1e-12
    mid = L.afo_compressor_compute_gain_reduction(comp, -20.0)
    assert abs(mid - 0.75 * 9.0 / 12.0) <= 1e-12


def test_pumping_score_properties(oracle):
    """python_api.rs:768-790."""
    L = oracle.lib()
    steady = np.full(250, 3.0, dtype=np.float32)
    assert L.afo_pumping_score(steady.ctypes.data_as(C.POINTER(C.c_float)), 250, 50.0) == 0.0
    i = np.arange(500, dtype=np.float32)
    fast = (3.0 + np.sin(2.0 * np.float32(np.pi) * 4.0 * i / 50.0)).astype(np.float32)
    slow = (3.0 + np.sin(2.0 * np.float32(np.pi) * 0.2 * i / 50.0)).astype(np.float32)
    f = L.afo_pumping_score(fast.ctypes.data_as(C.POINTER(C.c_float)), 500, 50.0)
    s = L.afo_pumping_score(slow.ctypes.data_as(C.POINTER(C.c_float)), 500, 50.0)
    assert f > 2.0 * s


LOOKAHEAD_PINS = {
    # evaluation/limiter-lookahead-report.json aggregates[ms]["controlled"]
    2.0: dict(gv=1.3907917598661823, te_med=-44.837684744690314, te_p90=-27.740017908805214),
    0.5: dict(gv=1.3906075587952735, te_med=-44.837684744690314, te_p90=-25.537891219961008),
}


@pytest.mark.parametrize("lookahead_ms", [2.0, 0.5])
def test_limiter_lookahead_report_pins(oracle, lookahead_ms):
    pins = LOOKAHEAD_PINS[lookahead_ms]
    gv, te, rows = [], [], []
    for _name, x in S.limiter_cases().items():
        r = oracle.simulate_auto_eq_chain(x, 48000, S.LIMITER_BANDS, S.limiter_settings(lookahead_ms))
        out = np.asarray(r["output_audio"], dtype=np.float64)
        delay = int(round(lookahead_ms / 1000.0 * 48000)) + 20
        aligned = out[delay:]
        ref = x[: aligned.size].astype(np.float64)
        gv.append(S.gain_envelope_variation_db(ref, aligned))
        te.append(S.transient_error_db(ref, aligned, S.transient_indices(ref)))
        rows.append(r)
        assert r["processed_samples"] == x.size and not r["non_finite_output"]
    ceiling = rows[0]["limiter_effective_ceiling_db"]
    assert ceiling == -1.5
    assert abs(float(np.median(gv)) - pins["gv"]) <= 1e-12
    assert abs(float(np.median(te)) - pins["te_med"]) <= 1e-9
    assert abs(float(np.percentile(te, 90.0)) - pins["te_p90"]) <= 1e-9
    assert sum(r["true_peak_limited_events"] for r in rows) == 1
    # dB stats pass through f32 log10 (Windows CRT vs glibc differ by <= 1 ulp of f32)
    assert abs(max(r["true_peak_limiter_gain_reduction_db"] for r in rows) - 0.5307239890098572) <= 2e-7
    assert abs(max(max(0.0, r["pre_limiter_true_peak_db"] - ceiling) for r in rows) - 0.5220339298248291) <= 2e-7
    assert max(max(0.0, r["output_true_peak_db"] - ceiling) for r in rows) == 0.0
    assert min(r["limiter_gain_reduction_db"] for r in rows) == 0.0


def test_dynamics_aliasing_report_pins(oracle):
    """evaluation/dynamics-aliasing-report.json cases[*].{base,reference}_peak_gain_reduction_db."""
    want = {
        ("carrier_8k", 48000): 16.455915451049805, ("carrier_8k", 192000): 17.137773513793945,
        ("carrier_11k", 48000): 16.86142921447754, ("carrier_11k", 192000): 17.001239776611328,
        ("carrier_15k", 48000): 16.1809024810791, ("carrier_15k", 192000): 16.64676856994629,
        ("carrier_18k", 48000): 15.484599113464355, ("carrier_18k", 192000): 16.35672378540039,
    }
    for name, carrier, mod in S.ALIASING_CASES:
        for fs in (48000, 192000):
            x = S.aliasing_signal(fs, carrier, mod)
            r = oracle.simulate_auto_eq_chain(x, fs, S.ALIASING_BANDS, S.ALIASING_SETTINGS)
            assert abs(r["compressor_gain_reduction_db"] - want[(name, fs)]) <= 1e-5 * want[(name, fs)]


def test_fir_table_matches_reference_text_if_present():
    """Only where the reference checkout exists (this container); skipped on the GPU box."""
    import pathlib
    import re

    src = pathlib.Path("/root/reference/rust-core/src/dsp/true_peak.rs")
    if not src.exists():
        pytest.skip("reference checkout not present")
    body = re.search(r"TRUE_PEAK_FIR[^=]*=\s*\[(.*?)\];\s*\n\s*#\[derive", src.read_text(), re.S).group(1)
    ref = np.array([float(v) for v in re.findall(r"-?\d+\.\d+(?:e-?\d+)?", body)]).reshape(4, 32).astype(np.float32)
    root = pathlib.Path(__file__).resolve().parents[1]
    for header in (root / "oracle" / "tp_fir_table.h", root / "audio-forge_amd" / "csrc" / "tp_fir_table.h"):
        vals = re.findall(r"-?0x[0-9a-f.]+p[-+]?\d+", header.read_text())
        mine = np.array([float.fromhex(v) for v in vals]).reshape(4, 32).astype(np.float32)
        assert np.array_equal(mine, ref)


def test_integrated_loudness_reference_properties(oracle):
    """loudness.rs:222-257 on the oracle's histogram-mode gate (ebur128 not vendored: unpinned)."""
    i = np.arange(48_000 * 8, dtype=np.float32)
    tone = (np.float32(0.1) * np.sin(np.float32(2.0 * np.pi) * np.float32(1000.0) * i / np.float32(48_000))).astype(np.float32)
    padded = np.concatenate([np.zeros(48_000, np.float32), tone, np.zeros(48_000, np.float32)])
    a, b = oracle.measure_integrated_loudness(tone, 48_000), oracle.measure_integrated_loudness(padded, 48_000)
    assert abs(a - b) < 0.2 and -24.0 < a < -22.0
    assert abs(a - (-23.05)) < 1e-9  # a block's loudness is its 0.1 LU histogram bin's centre
    for bad in (np.zeros(0, np.float32), np.array([np.nan], np.float32), np.zeros(48_000, np.float32)):
        with pytest.raises(ValueError):
            oracle.measure_integrated_loudness(bad, 48_000)
    with pytest.raises(ValueError):
        oracle.measure_integrated_loudness(np.array([0.1], np.float32), 12_345)


def test_gate_reference_properties(oracle):
    """gate.rs:958-1070 on the oracle's expander path."""
    g = oracle.Gate(-40.0, 10.0, 100.0)
    g.process(np.full(3_000, 0.1, np.float32))
    open_gain = g.current_gain
    assert open_gain > 0.8
    g.process(np.full(10_000, 0.0001, np.float32))
    assert g.current_gain < open_gain * 0.7 and g.current_gain < 0.5
    g = oracle.Gate(-40.0, 1.0, 1.0)
    g.process(np.zeros(4_000, np.float32))
    assert abs(g.current_gain - 10.0 ** (-36.0 / 20.0)) < 0.02
    g = oracle.Gate(-40.0, 1.0, 20.0)
    g.process(np.full(1, 0.1, np.float32))
    assert not g.is_open                                   # 8 ms RMS detector rejects a click
    g.process(np.full(2_000, 0.1, np.float32))
    assert g.is_open
    g.process(np.zeros(1_000, np.float32))
    assert g.is_open                                       # 50 ms hold
    g.process(np.zeros(4_000, np.float32))
    assert not g.is_open
    g = oracle.Gate(-40.0, 1.0, 10.0)
    for _ in range(5):
        g.process(np.full(2_000, 0.1, np.float32))
        g.process(np.zeros(4_500, np.float32))
    assert g.chatter_event_count > 0
