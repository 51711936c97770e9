"""The stage-pipeline form of the chain (audio-forge_amd/csrc/af_stages.hip, AF_KERNEL_STAGED) against the token-ring
kernel (AF_KERNEL_PHASED): the same expressions, operation for operation, in a build that does not contract
floating-point expressions, so audio and block rows must agree BIT FOR BIT -- whatever the oracle tolerances of the
libm-carrying stages are.  The oracle comparisons proper run in tests/test_gpu_parity.py (variant "staged")."""
import numpy as np
import pytest

import signals as S

pytestmark = pytest.mark.gpu

TYPED_BANDS = [("bell", 80.0 * 1.75**i, 3.0 if i % 2 else -2.5, 1.0, 12, True) for i in range(10)]


@pytest.fixture(scope="module")
def core():
    import mic_eq_mi
    from mic_eq_mi import _lib, mic_eq_core

    assert mic_eq_mi.CORE_AVAILABLE, "HIP library missing: GPU tests never fall back to the CPU"
    assert _lib.load().af_device_count() >= 1
    return mic_eq_core


LEGACY_BANDS = [(80.0 * 1.75**i, 3.0 if i % 2 else -2.5, 1.0) for i in range(10)]


def run(core, kernel, audio, settings, calls, bands=TYPED_BANDS, fs=48_000.0):
    eng = core.Engine(fs, audio.shape[0])
    try:
        settings = dict(settings)
        if len(bands[0]) == 3:  # the reference's (frequency, gain, q) setters: a 72-sample coefficient crossfade opens the stream
            core.configure_auto_eq_chain(eng, fs, bands, settings)
        else:
            settings["eq_bands_v2"] = bands  # typed bands: no crossfade
            core.configure_auto_eq_chain(eng, fs, S.LIMITER_BANDS, settings)
        eng.set_kernel(kernel)
        ys, rows = [], []
        for lo, hi in calls:
            ys.append(eng.process(audio[:, lo:hi]))
            rows.append(eng.block_stats().copy())
        assert eng._lib.af_engine_last_kernel(eng._h) == kernel
        return np.concatenate(ys, axis=1), np.concatenate(rows, axis=0)
    finally:
        eng.close()


def assert_same(a, b):
    ya, ra = a
    yb, rb = b
    assert ya.shape == yb.shape
    diff = np.flatnonzero(ya.view(np.uint32) != yb.view(np.uint32))
    assert diff.size == 0, f"{diff.size} samples differ, first at {np.unravel_index(diff[0], ya.shape) if ya.ndim > 1 else diff[0]}"
    for name in ra.dtype.names:
        assert np.array_equal(ra[name], rb[name]), name


CASES = {
    "full dynamics": {},
    "no side-chain filter": {"compressor_sidechain_highpass_enabled": False},
    "adaptive release": {"compressor_adaptive_release": True},
    "compressor only": {"limiter_enabled": False},
    "limiter only": {"compressor_enabled": False},
    "neither": {"compressor_enabled": False, "limiter_enabled": False},
    "short lookahead": {"limiter_lookahead_ms": 0.5},
    "auto makeup": {"compressor_auto_makeup_enabled": True, "compressor_target_lufs": -16.0},
    "auto makeup, adaptive": {"compressor_auto_makeup_enabled": True, "compressor_adaptive_release": True, "limiter_enabled": False},
    "long lookahead": {"limiter_lookahead_ms": 5.0},
}


@pytest.mark.parametrize("case", list(CASES))
def test_stage_pipeline_equals_token_ring(core, case):
    """70 streams (a full group and a ragged one), 2.1 s in three calls of unequal, partly ragged length: windows of ten
    control blocks, a short last window, a short last block, state and ring histories carried across calls."""
    from mic_eq_mi import _lib

    settings = dict(S.limiter_settings(2.0))
    settings.update(CASES[case])
    if case == "long lookahead":
        # 240 samples of lookahead do not fit the token ring's LDS layout: the 16-stream kernel is the yardstick
        ref_kernel = _lib.KERNEL_QUAD
    else:
        ref_kernel = _lib.KERNEL_PHASED
    audio = S.batch_signal(70, 210) * np.float32(1.6)  # loud enough for the limiter and the true-peak stage to act
    n = audio.shape[1]
    calls = ((0, 31_007), (31_007, 31_007 + 480 * 77), (31_007 + 480 * 77, n))
    want = run(core, ref_kernel, audio, settings, calls)
    got = run(core, _lib.KERNEL_STAGED, audio, settings, calls)
    assert_same(got, want)


@pytest.mark.parametrize("first_call", [31_007, 50, 72, 73, 5000])
def test_coefficient_crossfade_at_the_start(core, first_call):
    """The legacy band setters schedule a 72-sample crossfade per section (biquad.rs:263-327): the EQ stage runs its
    two-filter form while one is pending, also when the first call ends inside it."""
    from mic_eq_mi import _lib

    settings = dict(S.limiter_settings(2.0))
    audio = S.batch_signal(70, 70) * np.float32(1.6)
    n = audio.shape[1]
    calls = ((0, first_call), (first_call, n))
    want = run(core, _lib.KERNEL_PHASED, audio, settings, calls, bands=LEGACY_BANDS)
    got = run(core, _lib.KERNEL_STAGED, audio, settings, calls, bands=LEGACY_BANDS)
    assert_same(got, want)


def test_tiny_calls(core):
    """Calls shorter than a control block, shorter than the FIR history, and of one sample."""
    from mic_eq_mi import _lib

    settings = dict(S.limiter_settings(2.0))
    audio = S.batch_signal(5, 12) * np.float32(1.5)
    edges = [0, 1, 2, 33, 100, 479, 480, 481, 1500, 1501, 4000, audio.shape[1]]
    calls = list(zip(edges[:-1], edges[1:]))
    want = run(core, _lib.KERNEL_PHASED, audio, settings, calls)
    got = run(core, _lib.KERNEL_STAGED, audio, settings, calls)
    assert_same(got, want)


def test_non_finite_input_and_scrub(core):
    """NaN / Inf samples: the scrub in front, the limiter's and the detector's own scrubs, the non-finite flag."""
    from mic_eq_mi import _lib

    settings = dict(S.limiter_settings(2.0))
    audio = S.batch_signal(3, 20).copy()
    audio[0, 1000] = np.nan
    audio[1, 2000:2004] = np.inf
    audio[2, 3000] = -np.inf
    for extra in ({}, {"compressor_enabled": False, "limiter_enabled": False}):
        s = dict(settings)
        s.update(extra)
        calls = ((0, audio.shape[1]),)
        want = run(core, _lib.KERNEL_PHASED, audio, s, calls)
        got = run(core, _lib.KERNEL_STAGED, audio, s, calls)
        ya, yb = got[0], want[0]
        assert np.array_equal(np.isnan(ya), np.isnan(yb))
        assert_same((np.nan_to_num(ya, nan=7.0), got[1]), (np.nan_to_num(yb, nan=7.0), want[1]))


def test_unsupported_configurations_are_refused(core):
    from mic_eq_mi import _lib

    eng = core.Engine(48_000.0, 2)
    try:
        settings = dict(S.limiter_settings(2.0))
        settings["eq_bands_v2"] = TYPED_BANDS
        core.configure_auto_eq_chain(eng, 48_000.0, S.LIMITER_BANDS, settings)
        eng.set_deesser_enabled(1)
        eng.set_eq_before_deesser(1)  # (the de-esser ahead of the EQ is built as stages since round 3; this order is not)
        eng.set_kernel(_lib.KERNEL_STAGED)
        with pytest.raises(NotImplementedError, match="stage pipeline"):
            eng.process(S.batch_signal(2, 2))
        eng.reset()
        eng.set_eq_before_deesser(0)
        eng.process(S.batch_signal(2, 2))
        assert eng.last_kernel() == _lib.KERNEL_STAGED
    finally:
        eng.close()


def test_randomized_configurations_against_the_other_kernels(core):
    """Seeded random chain settings, batch shapes and call patterns: the stage pipeline against kernel 2 (kernel 3 where the
    lookahead does not fit kernel 2's LDS layout), bit for bit.  What a hand-picked list of cases would miss."""
    from mic_eq_mi import _lib

    import os

    rng = np.random.default_rng(int(os.environ.get("AF_RANDOM_SEED", "20261004")))
    kinds = ("bell", "low_shelf", "high_shelf", "high_pass", "low_pass", "notch")
    for case in range(int(os.environ.get("AF_RANDOM_CASES", "14"))):  # (a soak run: AF_RANDOM_CASES=200 AF_RANDOM_SEED=...)
        fs = float(rng.choice([44_100.0, 48_000.0, 48_000.0, 96_000.0]))
        n_streams = int(rng.choice([1, 3, 64, 65, 130]))
        blocks = int(rng.integers(8, 60))
        audio = (S.batch_signal(n_streams, blocks) * np.float32(rng.uniform(0.3, 2.5))).astype(np.float32)
        n = audio.shape[1]
        bands = []
        for i in range(10):
            kind = kinds[int(rng.integers(0, len(kinds)))]
            slope = int(rng.choice([12, 12, 24])) if kind in ("high_pass", "low_pass") else 12
            bands.append((kind, float(60.0 * 1.8**i * rng.uniform(0.8, 1.2)), float(rng.uniform(-9.0, 9.0)), float(rng.uniform(0.5, 3.0)), slope,
                          bool(rng.random() > 0.15)))
        lookahead = float(rng.choice([0.25, 0.5, 1.0, 2.0, 2.0, 3.0]))
        settings = dict(S.limiter_settings(lookahead))
        settings.update({
            "compressor_enabled": bool(rng.random() > 0.2),
            "compressor_threshold_db": float(rng.uniform(-40.0, -8.0)),
            "compressor_ratio": float(rng.uniform(1.5, 10.0)),
            "compressor_attack_ms": float(rng.uniform(0.5, 40.0)),
            "compressor_release_ms": float(rng.uniform(20.0, 500.0)),
            "compressor_makeup_gain_db": float(rng.uniform(0.0, 9.0)),
            "compressor_adaptive_release": bool(rng.random() > 0.5),
            "compressor_sidechain_highpass_enabled": bool(rng.random() > 0.3),
            "limiter_enabled": bool(rng.random() > 0.2),
            "limiter_ceiling_db": float(rng.uniform(-6.0, -0.1)),
            "limiter_release_ms": float(rng.uniform(10.0, 200.0)),
        })
        cuts = sorted(set(int(v) for v in rng.integers(1, n, size=int(rng.integers(0, 4)))))
        edges = [0, *cuts, n]
        calls = list(zip(edges[:-1], edges[1:]))
        lookahead_samples = round(lookahead * fs / 1000.0)
        ref_kernel = _lib.KERNEL_PHASED if lookahead_samples <= 120 else _lib.KERNEL_QUAD
        if ref_kernel == _lib.KERNEL_PHASED and settings["compressor_enabled"] and rng.random() > 0.6:
            settings["compressor_auto_makeup_enabled"] = True  # (only kernel 2 has it to compare with)
            settings["compressor_target_lufs"] = float(rng.uniform(-24.0, -12.0))
        use_legacy = bool(rng.random() > 0.6)
        legacy = [(b[1], b[2], b[3]) for b in bands]
        kw = dict(bands=legacy if use_legacy else bands, fs=fs)
        try:
            want = run(core, ref_kernel, audio, settings, calls, **kw)
        except NotImplementedError:  # kernel 2's LDS layout does not hold this EQ / lookahead: kernel 3 is the yardstick
            settings["compressor_auto_makeup_enabled"] = False
            want = run(core, _lib.KERNEL_QUAD, audio, settings, calls, **kw)
        got = run(core, _lib.KERNEL_STAGED, audio, settings, calls, **kw)
        try:
            assert_same(got, want)
        except AssertionError as err:
            raise AssertionError(f"case {case}: fs {fs}, {n_streams} streams, calls {calls}, settings {settings}: {err}") from None


def test_several_presets_in_one_engine(core):
    """Three presets over 150 streams (one per 64-stream group) that agree on which stages run and differ in everything else
    (EQ layout incl. section counts, thresholds, time constants, ceilings, lookahead): the stage pipeline reads each group's
    own parameter block like kernel 2 does -- same bits, two calls."""
    import ctypes as C

    from mic_eq_mi import _lib

    steep = list(S.DEFAULT_TYPED_BANDS)
    steep[0] = ("high_pass", 90.0, 0.0, 0.707, 36, True)
    steep[4] = ("bell", 1000.0, 6.0, 2.0, 12, True)
    presets = [
        (LEGACY_BANDS, dict(S.limiter_settings(2.0))),
        (S.LIMITER_BANDS, dict(S.limiter_settings(1.0), compressor_threshold_db=-30.0, compressor_ratio=2.0, compressor_release_ms=90.0,
                               limiter_ceiling_db=-3.0, limiter_careful_output_enabled=False)),
        (S.LIMITER_BANDS, dict(S.limiter_settings(0.5), compressor_makeup_gain_db=6.0, compressor_attack_ms=2.0, eq_bands_v2=steep)),
    ]
    audio = S.batch_signal(150, 40) * np.float32(1.7)
    outs = {}
    for kernel in (_lib.KERNEL_PHASED, _lib.KERNEL_STAGED):
        eng = core.Engine(48_000.0, 150)
        try:
            eng.set_preset_count(3)
            for k, (bands, settings) in enumerate(presets):
                eng.select_preset(k)
                core.configure_auto_eq_chain(eng, 48_000.0, bands, settings)
            gp = np.asarray([2, 0, 1], dtype=np.int32)
            _lib.check(eng._lib.af_engine_assign_presets(eng._h, gp.ctypes.data_as(C.POINTER(C.c_int32)), 3))
            eng.set_kernel(kernel)
            ys, rows = [], []
            for lo, hi in ((0, 7001), (7001, audio.shape[1])):
                ys.append(eng.process(audio[:, lo:hi]))
                rows.append(eng.block_stats().copy())
            assert eng._lib.af_engine_last_kernel(eng._h) == kernel
            outs[kernel] = (np.concatenate(ys, axis=1), np.concatenate(rows, axis=0))
        finally:
            eng.close()
    assert_same(outs[_lib.KERNEL_STAGED], outs[_lib.KERNEL_PHASED])


def test_engine_reuse_after_reset_with_shorter_control_blocks(core):
    """An engine that ran the pipeline, is reset, reconfigured with shorter control blocks (more rows per window in the
    per-block arrays) and run again gives what a fresh engine gives."""
    from mic_eq_mi import _lib

    audio = S.batch_signal(70, 30) * np.float32(1.5)
    settings = dict(S.limiter_settings(2.0), compressor_auto_makeup_enabled=True)
    settings["eq_bands_v2"] = TYPED_BANDS

    def second_config(eng):
        core.configure_auto_eq_chain(eng, 48_000.0, S.LIMITER_BANDS, dict(settings, compressor_threshold_db=-28.0))
        eng.set_control_block_samples(480)
        eng.set_kernel(_lib.KERNEL_STAGED)

    eng = core.Engine(48_000.0, 70)
    try:
        core.configure_auto_eq_chain(eng, 48_000.0, S.LIMITER_BANDS, settings)
        eng.set_kernel(_lib.KERNEL_STAGED)
        eng.process(audio)
        eng.reset()
        second_config(eng)
        got = (eng.process(audio), eng.block_stats().copy())
    finally:
        eng.close()
    fresh = core.Engine(48_000.0, 70)
    try:
        second_config(fresh)
        want = (fresh.process(audio), fresh.block_stats().copy())
    finally:
        fresh.close()
    assert_same(got, want)


def test_baseline_configs1_shape_on_auto(oracle):
    """BASELINE configs[1] at its own shape -- 256 streams x 10 s, EQ + compressor + limiter + true-peak limiter, no suppressor --
    on AUTO (which routes it to the stage pipeline): sixteen streams spread over the batch against the CPU oracle, determinism,
    energy bookkeeping of the block rows against the audio."""
    import torch

    import bench
    from mic_eq_mi import mic_eq_core as core

    streams, seconds = 256, 10
    dev = torch.device("cuda", 0)
    x = bench.synth_batch(streams, seconds * 100, 0, dev)
    n = x.shape[1]

    def run():
        y = torch.empty_like(x)
        eng = core.Engine(48_000.0, streams, 0)
        core.configure_auto_eq_chain(eng, 48_000.0, bench.BANDS, bench.CHAIN_SETTINGS)
        eng.process_device(x.data_ptr(), y.data_ptr(), n, n, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        rows, used = eng.block_stats(), eng.last_kernel()
        eng.close()
        return y, rows, used

    y, rows, used = run()
    assert used == 4  # AF_KERNEL_STAGED
    y2, _, _ = run()
    assert torch.equal(y, y2)
    assert rows.shape == (seconds * 50, streams)
    assert np.allclose(rows["output_square_sum"].sum(axis=0), (y.double() ** 2).sum(dim=1).cpu().numpy(), rtol=1e-9)
    sample = sorted({0, 1, 63, 64, 127, 128, 255} | {int(v) for v in np.linspace(0, streams - 1, 12)})
    assert len(sample) >= 16
    worst = (0.0, 0.0)
    for s in sample:
        want = oracle.simulate_auto_eq_chain(x[s].cpu().numpy(), 48_000, bench.BANDS, dict(bench.CHAIN_SETTINGS, return_output_audio=True))["output_audio"]
        d = y[s].cpu().numpy().astype(np.float64) - want.astype(np.float64)
        worst = (max(worst[0], float(np.max(np.abs(d)))), max(worst[1], float(np.sqrt(np.mean(d * d)))))
    print(f"configs[1] on AUTO (stage pipeline): {len(sample)} streams vs oracle, worst |err| {worst[0]:.3e}, worst RMS {worst[1]:.3e}")
    assert worst[0] <= 2e-7 and worst[1] <= 2e-8, worst
