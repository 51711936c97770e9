"""CPU checks of the drop-in boundary: the C ABI library loads and exports every symbol the
header declares, and the host-only entry points (configuration, EQ response) behave like the
reference's operators.  No GPU compute is touched here."""
import ctypes as C
import pathlib
import re

import numpy as np
import pytest

import signals as S

ROOT = pathlib.Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def mi():
    import mic_eq_mi

    assert mic_eq_mi.CORE_AVAILABLE, "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    return mic_eq_mi


def test_library_exports_every_declared_symbol(mi):
    from mic_eq_mi import _lib

    header = (ROOT / "include" / "audioforge_mi.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(af_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 55
    lib = C.CDLL(str(_lib.LIB_PATH))
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, missing
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.af_version() >= 100


def test_package_imports_under_its_hyphenated_name():
    import importlib

    pkg = importlib.import_module("audio-forge_amd")
    assert pkg.mic_eq_mi.CORE_AVAILABLE
    assert callable(pkg.simulate_auto_eq_chain)


def test_operator_table_is_complete_and_missing_core_raises_import_error(mi):
    """python/mic_eq/__init__.py:38-73: every offline operator name the reference re-exports resolves to a
    built function; the degradation path for an absent one is the reference's own `_missing_core` (ImportError)."""
    for name in ("simulate_auto_eq_chain", "simulate_auto_makeup_control", "simulate_gate_suppressor_order",
                 "simulate_eq_v2", "simulate_product_resampler", "product_resampler_configuration",
                 "eq_magnitude_response", "eq_magnitude_response_v2", "measure_integrated_loudness"):
        fn = getattr(mi, name)
        assert callable(fn) and fn is not mi._missing_core, name
    with pytest.raises(ImportError):
        mi._missing_core(np.zeros(4, dtype=np.float32))
    with pytest.raises(ValueError):  # argument contract is checked before any GPU work (python_api.rs:296-311)
        mi.simulate_gate_suppressor_order(np.zeros(4, dtype=np.float32), [0.0, 0.0], True)


def test_eq_response_matches_oracle_bit_for_bit(mi, oracle):
    grid = np.geomspace(20.0, 20_000.0, 512)
    got = np.asarray(mi.eq_magnitude_response_v2(grid.tolist(), S.DEFAULT_TYPED_BANDS, 48_000.0))
    assert np.array_equal(got, oracle.eq_magnitude_response_v2(grid, S.DEFAULT_TYPED_BANDS, 48_000.0))
    legacy = [(80.0 * 1.6**k, (-1) ** k * 2.5, 0.7 + 0.1 * k) for k in range(10)]
    got = np.asarray(mi.eq_magnitude_response(grid.tolist(), legacy, 48_000.0))
    assert np.array_equal(got, oracle.eq_magnitude_response(grid, legacy, 48_000.0))
    bands = list(S.DEFAULT_TYPED_BANDS)
    bands[3] = ("high_pass", 300.0, 0.0, 1.0, 48, True)
    bands[7] = ("notch", 5000.0, 3.0, 6.0, 12, True)
    bands[8] = ("low_pass", 9000.0, 0.0, 1.0, 36, False)
    got = np.asarray(mi.eq_magnitude_response_v2(grid.tolist(), bands, 48_000.0))
    assert np.array_equal(got, oracle.eq_magnitude_response_v2(grid, bands, 48_000.0))


def test_eq_response_error_contract(mi):
    """python/tests/test_eq_native_response.py:26-69 and test_eq_filter_types.py:50-56 substrings."""
    grid = [100.0, 1000.0]
    with pytest.raises(ValueError, match="expected 10 EQ bands"):
        mi.eq_magnitude_response(grid, [(100.0, 0.0, 1.0)] * 9, 48_000.0)
    with pytest.raises(ValueError, match="sample_rate"):
        mi.eq_magnitude_response(grid, [(100.0, 0.0, 1.0)] * 10, 0.0)
    with pytest.raises(ValueError, match="Nyquist"):
        mi.eq_magnitude_response(grid, [(30_000.0, 0.0, 1.0)] * 10, 48_000.0)
    with pytest.raises(ValueError, match="Nyquist"):
        mi.eq_magnitude_response([30_000.0], [(100.0, 0.0, 1.0)] * 10, 48_000.0)
    bad = list(S.DEFAULT_TYPED_BANDS)
    bad[2] = ("tilt", 300.0, 0.0, 1.0, 12, True)
    with pytest.raises(ValueError, match="unsupported EQ filter type"):
        mi.eq_magnitude_response_v2(grid, bad, 48_000.0)
    bad[2] = ("high_pass", 300.0, 0.0, 1.0, 18, True)
    with pytest.raises(ValueError, match="expected one of"):
        mi.eq_magnitude_response_v2(grid, bad, 48_000.0)
    bad[2] = ("bell", 300.0, 13.0, 1.0, 12, True)
    with pytest.raises(ValueError, match="gain"):
        mi.eq_magnitude_response_v2(grid, bad, 48_000.0)


def test_engine_configuration_is_host_only(mi):
    """Creating and configuring an engine needs no device; setters validate like the reference."""
    e = mi.Engine(48_000.0, 8)
    e.set_compressor_enabled(1)
    e.compressor_set_threshold(-22.0)
    e.limiter_set_lookahead_ms(2.0)
    assert e.limiter_lookahead_samples() == 96
    e.limiter_set_ceiling(3.0)  # limiter.rs:139-142 clamps to <= 0 dB
    assert e.limiter_ceiling_db() == 0.0
    with pytest.raises(ValueError):
        e.eq_set_band_gain(10, 1.0)
    with pytest.raises(ValueError):
        e.set_control_block_samples(0)
    e.close()
    with pytest.raises(ValueError, match="sample_rate"):
        mi.Engine(float("nan"), 1)


def test_simulate_argument_contract(mi):
    audio = np.zeros(16, dtype=np.float32)
    with pytest.raises(ValueError, match="expected 10 EQ bands"):
        mi.simulate_auto_eq_chain(audio, 48_000, [(100.0, 0.0, 1.0)] * 3)
    with pytest.raises(ValueError, match="sample_rate"):
        mi.simulate_auto_eq_chain(audio, -1.0, S.LIMITER_BANDS)
    with pytest.raises(TypeError):
        mi.simulate_auto_eq_chain(audio.astype(np.float64), 48_000, S.LIMITER_BANDS)
    with pytest.raises(ValueError, match="contiguous"):
        mi.simulate_auto_eq_chain(np.zeros(32, dtype=np.float32)[::2], 48_000, S.LIMITER_BANDS)
    nan_audio = audio.copy()
    nan_audio[3] = np.nan
    with pytest.raises(ValueError, match="audio must contain only finite samples"):
        mi.simulate_eq_v2(nan_audio, 48_000.0, S.DEFAULT_TYPED_BANDS)


def test_chain_diagnostics_host_logic_matches_oracle(oracle):
    """The dict statistics (python_api.rs:578-713) are host logic: feed them oracle rows."""
    from mic_eq_mi import mic_eq_core as core

    L = oracle.lib()
    x = S.limiter_cases()["controlled-clipped-voice"]
    settings = S.limiter_settings(2.0)
    want = oracle.simulate_auto_eq_chain(x, 48_000, S.LIMITER_BANDS, settings)
    # rebuild the per-block rows with the oracle's block processor
    chain = oracle.Chain(48_000.0)
    for k, (f, g, q) in enumerate(S.LIMITER_BANDS):
        L.afo_eq_set_band_frequency(chain.eq, k, f)
        L.afo_eq_set_band_gain(chain.eq, k, g)
        L.afo_eq_set_band_q(chain.eq, k, q)
    chain.set("compressor_enabled", 1)
    c = chain.compressor
    L.afo_compressor_set_threshold(c, -20.0); L.afo_compressor_set_ratio(c, 4.0)
    L.afo_compressor_set_attack_time(c, 10.0); L.afo_compressor_set_release_time(c, 200.0)
    L.afo_compressor_set_makeup_gain(c, 0.0); L.afo_compressor_set_adaptive_release(c, 0)
    L.afo_compressor_set_base_release_time(c, 50.0); L.afo_compressor_set_auto_makeup_enabled(c, 0)
    L.afo_compressor_set_target_lufs(c, -18.0); L.afo_compressor_set_sidechain_highpass_enabled(c, 1)
    L.afo_limiter_set_lookahead_ms(chain.limiter, 2.0); L.afo_limiter_set_ceiling(chain.limiter, -1.5)
    L.afo_limiter_set_release_time(chain.limiter, 50.0); L.afo_tp_limiter_set_release_ms(chain.tp_limiter, 50.0)
    y = x.copy()
    n_blocks = (x.size + 959) // 960
    rows = np.zeros(n_blocks, dtype=core.STATS_DTYPE)
    lengths = core._block_lengths(x.size, 960)
    for b in range(n_blocks):
        blk = y[b * 960 : (b + 1) * 960]
        rows["input_square_sum"][b] = float(np.sum(blk.astype(np.float64) ** 2))
        st = chain.process_block(blk)
        rows["output_square_sum"][b] = float(np.sum(blk.astype(np.float64) ** 2))
        for name in ("input_sample_peak", "output_sample_peak", "true_peak_limiter_input_peak", "output_true_peak",
                     "limiter_peak_gain_reduction_db", "true_peak_limiter_gain_reduction_db",
                     "compressor_gain_reduction_db", "deesser_gain_reduction_db", "true_peak_limited_events"):
            rows[name][b] = getattr(st, name)
    got = core.chain_diagnostics(rows, lengths, -1.5)
    assert np.array_equal(y, want["output_audio"])
    for key, value in got.items():
        ref = want[key]
        if isinstance(value, (bool, int)):
            assert value == ref, key
        else:
            assert abs(value - ref) <= 2e-5 * max(1.0, abs(ref)), (key, value, ref)


def test_resampler_host_plan_matches_oracle(oracle):
    """The resampler's host side (coefficient table, chunk-loop replay, argument contract) needs no GPU:
    table bit-identical to the oracle's, frame/chunk counts identical for ragged lengths and ratios."""
    import ctypes as C

    from mic_eq_mi import mic_eq_core as core

    L = oracle.lib()
    L.afo_resampler_new.restype = C.c_void_p
    L.afo_resampler_new.argtypes = [C.c_uint32, C.c_uint32, C.c_size_t, C.c_size_t, C.c_int, C.c_float]
    L.afo_resampler_sinc_table.restype = C.POINTER(C.c_double)
    L.afo_resampler_sinc_table.argtypes = [C.c_void_p]
    L.afo_resampler_free.argtypes = [C.c_void_p]
    for fi, fo, sinc_len, window in ((44_100, 48_000, 128, "blackman"), (48_000, 44_100, 128, "blackman"),
                                     (48_000, 44_100, 256, "blackman_harris_squared"), (32_000, 48_000, 64, "hann_squared")):
        r = core.Resampler(fi, fo, 1024, sinc_len, window)
        h = L.afo_resampler_new(fi, fo, 1024, sinc_len, oracle.RESAMPLER_WINDOWS[window], 0.0)
        table = np.ctypeslib.as_array(L.afo_resampler_sinc_table(h), shape=(256, sinc_len))
        assert np.array_equal(r.sinc_table(), table)
        L.afo_resampler_free(h)
        for n in (0, 1, 1023, 1024, 1025, 44_100, 100_003):
            y, delay, expected, blocks = oracle.simulate_product_resampler(np.zeros(n), fi, fo, 1024, sinc_len, window)
            assert r.plan(n) == (y.size, blocks), (fi, fo, n)
            assert (r.output_delay, r.expected_frames(n)) == (delay, expected)
        r.close()
    assert core.product_resampler_configuration() == (128, "blackman", "cubic", 256, 1024)  # tests.rs:209-221
    for bad in ((0, 48_000, 1024, None, None), (48_000, 44_100, 0, None, None), (48_000, 44_100, 1025, None, None),
                (48_000, 44_100, 1024, 96, None), (48_000, 44_100, 1024, 4096, None), (48_000, 44_100, 1024, None, "unknown")):
        with pytest.raises(ValueError):
            core.Resampler(*bad)
