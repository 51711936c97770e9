"""Several presets in one engine (one per 64-stream group): the reference configures one processor per stream
(python/mic_eq/config_parts/settings.py:543-593), so a batch whose streams carry different EQ / compressor / limiter
settings must equal per-stream single-preset runs of the oracle -- bit for bit where no device libm call is involved,
within 2e-7 with the compressor (the tolerance of tests/test_gpu_parity.py)."""
import numpy as np
import pytest

import signals as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mi():
    import mic_eq_mi

    assert mic_eq_mi.CORE_AVAILABLE
    return mic_eq_mi


def _presets():
    lim_only = dict(S.limiter_settings(2.0), compressor_enabled=False)
    bands_a = [(80.0 * 1.75**i, 3.0 if i % 2 else -2.5, 1.0) for i in range(10)]
    comp = S.limiter_settings(2.0)
    bands_b = S.LIMITER_BANDS
    steep = list(S.DEFAULT_TYPED_BANDS)
    steep[0] = ("high_pass", 90.0, 0.0, 0.707, 48, True)   # four sections
    steep[9] = ("low_pass", 15000.0, 0.0, 0.707, 36, True)  # three sections
    steep[4] = ("bell", 1000.0, 6.0, 2.0, 12, True)
    other = dict(S.limiter_settings(1.0), compressor_adaptive_release=True, compressor_threshold_db=-26.0, compressor_ratio=2.5,
                 limiter_ceiling_db=-3.0, limiter_careful_output_enabled=False, eq_bands_v2=steep)
    return [(bands_a, lim_only), (bands_b, comp), (S.LIMITER_BANDS, other)]


def test_three_presets_in_one_batch(mi, oracle):
    audio = S.batch_signal(150, 60)  # 0.6 s
    presets = _presets()
    which = [(7 * i + i // 5) % 3 for i in range(audio.shape[0])]  # interleaved: every preset's streams are scattered
    out, results = mi.simulate_auto_eq_chain_batch(audio, 48_000.0, [presets[k][0] for k in which], [presets[k][1] for k in which])
    assert out.shape == audio.shape and len(results) == audio.shape[0]
    checked = {0: 0, 1: 0, 2: 0}
    for s in list(range(0, 150, 11)) + [63, 64, 127, 128, 149]:
        k = which[s]
        bands, settings = presets[k]
        want = oracle.simulate_auto_eq_chain(audio[s], 48_000, bands, dict(settings))
        ref = np.asarray(want["output_audio"], dtype=np.float32)
        if k == 0:  # EQ + limiter + true-peak limiter: no device libm call
            assert np.array_equal(out[s].view(np.uint32), ref.view(np.uint32)), s
        else:
            assert float(np.max(np.abs(out[s].astype(np.float64) - ref.astype(np.float64)))) <= 2e-7, (s, k)
        for key in ("true_peak_limited_events", "processed_samples"):
            assert results[s][key] == want[key], (s, key)
        assert abs(results[s]["limiter_effective_ceiling_db"] - want["limiter_effective_ceiling_db"]) < 1e-6
        checked[k] += 1
    assert all(v >= 3 for v in checked.values())


def test_presets_behind_the_suppressor_and_across_calls(mi, oracle):
    """Two presets, the realtime front end and the suppressor ahead of them, two calls (state and crossfade bookkeeping per
    preset): group 0 (64 streams) runs preset 1, group 1 (6 streams) preset 0."""
    from mic_eq_mi import mic_eq_core as core
    import ctypes as C

    audio = S.batch_signal(70, 120)
    presets = _presets()[:2]
    eng = core.Engine(48_000.0, 70)
    eng.set_preset_count(2)
    for k, (bands, settings) in enumerate(presets):
        eng.select_preset(k)
        core.configure_auto_eq_chain(eng, 48_000.0, bands, settings)
    assert eng.preset_count() == 2
    gp = np.asarray([1, 0], dtype=np.int32)
    core._lib.check(eng._lib.af_engine_assign_presets(eng._h, gp.ctypes.data_as(C.POINTER(C.c_int32)), 2))
    eng.set_prefilter_enabled(1, 1)
    eng.set_suppressor_enabled(1)
    a = eng.process(audio[:, : 50 * 480])
    b = eng.process(audio[:, 50 * 480 :])
    eng.close()
    got = np.concatenate([a, b], axis=1)
    for s, k in ((0, 1), (63, 1), (64, 0), (69, 0)):
        bands, settings = presets[k]
        sup = oracle.suppressor_process(oracle.prefilter(audio[s]), 1.0)
        want = np.asarray(oracle.simulate_auto_eq_chain(sup, 48_000, bands, dict(settings))["output_audio"], dtype=np.float64)
        d = got[s].astype(np.float64) - want
        assert float(np.sqrt(np.mean(d * d))) <= 1e-5, (s, k)


def test_preset_argument_contract(mi):
    from mic_eq_mi import mic_eq_core as core
    import ctypes as C

    eng = core.Engine(48_000.0, 130)
    with pytest.raises(ValueError):
        eng.set_preset_count(0)
    eng.set_preset_count(3)
    with pytest.raises(ValueError):
        eng.select_preset(3)
    gp = np.asarray([0, 1], dtype=np.int32)  # 130 streams are three groups
    with pytest.raises(ValueError, match="one preset index per group"):
        core._lib.check(eng._lib.af_engine_assign_presets(eng._h, gp.ctypes.data_as(C.POINTER(C.c_int32)), 2))
    gp = np.asarray([0, 1, 5], dtype=np.int32)
    with pytest.raises(ValueError, match="does not exist"):
        core._lib.check(eng._lib.af_engine_assign_presets(eng._h, gp.ctypes.data_as(C.POINTER(C.c_int32)), 3))
    eng.select_preset(1)
    eng.set_deesser_enabled(1)  # ONE preset with the de-esser (the others without): no kernel runs groups with different stages
    with pytest.raises(NotImplementedError, match="every preset must enable the same stages"):
        eng.process(np.zeros((130, 960), dtype=np.float32))
    eng.close()


@pytest.mark.parametrize("deesser", [False, True], ids=["automakeup", "deesser+automakeup"])
def test_presets_with_auto_makeup_and_the_deesser(mi, oracle, deesser):
    """Three presets that all run the compressor's auto-makeup (and, second case, the de-esser ahead of the EQ), different in
    every coefficient: the stage pipeline runs them as one batch (each 64-stream group reads its own parameter block, per
    window for the de-esser's crossfade bookkeeping).  Equal to per-stream single-preset runs of the oracle."""
    audio = S.batch_signal(200, 120)  # 1.2 s: the makeup gain has left its initial value
    audio[:, 20_000:30_000] *= 3.0
    presets = []
    for k, (bands, settings) in enumerate(_presets()):
        st = dict(settings, compressor_enabled=True, compressor_auto_makeup_enabled=True, compressor_target_lufs=-18.0 - 2.0 * k,
                  compressor_threshold_db=-30.0 + 3.0 * k, compressor_ratio=3.0 + k, compressor_adaptive_release=True,
                  compressor_base_release_ms=120.0 + 40.0 * k)  # (which stages run must agree: adaptive release on all three)
        if deesser:
            st.update(deesser_enabled=True, deesser_auto_enabled=bool(k != 1), deesser_auto_amount=0.4 + 0.2 * k,
                      deesser_threshold_db=-42.0 + 2.0 * k, deesser_ratio=5.0, deesser_low_cut_hz=3500.0 + 500.0 * k,
                      deesser_high_cut_hz=9000.0 + 800.0 * k, deesser_max_reduction_db=6.0 + 2.0 * k)
        presets.append((bands, st))
    which = [(5 * i + i // 7) % 3 for i in range(audio.shape[0])]
    out, results = mi.simulate_auto_eq_chain_batch(audio, 48_000.0, [presets[k][0] for k in which], [presets[k][1] for k in which])
    checked = {0: 0, 1: 0, 2: 0}
    worst = 0.0
    for s in list(range(0, 200, 17)) + [63, 64, 127, 128, 191, 192, 199]:
        k = which[s]
        bands, settings = presets[k]
        want = oracle.simulate_auto_eq_chain(audio[s], 48_000, bands, dict(settings, return_output_audio=True))
        ref = np.asarray(want["output_audio"], dtype=np.float64)
        err = float(np.max(np.abs(out[s].astype(np.float64) - ref)))
        worst = max(worst, err)
        assert err <= 2e-7, (s, k, err)
        assert results[s]["processed_samples"] == want["processed_samples"]
        checked[k] += 1
    assert all(v >= 3 for v in checked.values())
    assert float(np.max(np.abs(out))) > 0.05  # (something came out)
