"""GPU parity of the batched product resampler with the CPU oracle (oracle/af_resampler.c), plus the
reference's published measurements (tests/golden/resampler_report_pins.json) through the HIP path.

The kernel and the oracle evaluate every sinc dot product as the same fused multiply-add chain and the same
cubic, on the same host-built coefficient table and positions: outputs must be BIT-EXACT (f64).
"""
import json
import pathlib

import numpy as np
import pytest

import resampler_stimuli as R

pytestmark = pytest.mark.gpu

PINS = json.loads((pathlib.Path(__file__).parent / "golden" / "resampler_report_pins.json").read_text())


@pytest.fixture(scope="module")
def mi():
    import mic_eq_mi

    assert mic_eq_mi.CORE_AVAILABLE, "HIP library missing: GPU tests never fall back to the CPU"
    return mic_eq_mi


@pytest.mark.parametrize("fi,fo", [(44_100, 48_000), (48_000, 44_100), (32_000, 48_000), (96_000, 48_000), (48_000, 16_000)])
def test_bit_exact_against_oracle(mi, oracle, fi, fo):
    """Ragged batch (67 streams), length that is not a multiple of the chunk, several ratios incl. the
    segment sizes 128 / 64 / 32 of the kernel."""
    from mic_eq_mi import mic_eq_core as core

    rng = np.random.default_rng(fi ^ fo)
    n = 3 * 1024 + 517
    x = rng.standard_normal((67, n)) * 0.25
    x[3] = 0.0
    x[5, 1000] = 1.0
    out, delay, expected, blocks, _ = core.simulate_product_resampler_batch(x, fi, fo)
    for s in (0, 3, 5, 63, 64, 66):
        want, d, e, b = oracle.simulate_product_resampler(x[s], fi, fo)
        assert (delay, expected, blocks) == (d, e, b)
        assert out.shape[1] == want.size
        assert np.array_equal(out[s], want), (s, float(np.max(np.abs(out[s] - want))))


def test_operator_surface_matches_reference(mi, oracle):
    """tests.rs:194-257: tuple shape, expected = 48000, delay, len >= delay + expected, timings non-empty,
    configuration tuple, and the ValueError contract."""
    out, delay, expected, timings = mi.simulate_product_resampler([0.0] * 44_100, 44_100, 48_000, 1024, None, None)
    assert expected == 48_000 and delay == 69 and len(out) >= delay + expected and len(timings) == 44
    assert isinstance(out, list) and isinstance(out[0], float) and all(isinstance(t, int) for t in timings)
    assert mi.product_resampler_configuration() == (128, "blackman", "cubic", 256, 1024)
    for args in (([0.0], 0, 48_000, 1024, None, None), ([float("nan")], 48_000, 44_100, 1024, None, None),
                 ([0.0], 48_000, 44_100, 0, None, None), ([0.0], 48_000, 44_100, 1025, None, None),
                 ([0.0], 48_000, 44_100, 1024, 96, None), ([0.0], 48_000, 44_100, 1024, None, "unknown")):
        with pytest.raises(ValueError):
            mi.simulate_product_resampler(*args)
    # other windows / a longer sinc run too and agree with the oracle
    x = np.sin(np.arange(5000) * 0.05)
    for sinc_len, window in ((256, "blackman_harris_squared"), (64, "hann"), (128, "blackman_squared")):
        got = np.asarray(mi.simulate_product_resampler(x.tolist(), 48_000, 44_100, 1024, sinc_len, window)[0])
        want = oracle.simulate_product_resampler(x, 48_000, 44_100, 1024, sinc_len, window)[0]
        assert np.array_equal(got, want), (sinc_len, window)


def test_published_measurements_through_the_gpu(mi):
    """evaluation/resampler-quality-report.json figures, computed from the HIP path's output."""
    pins = PINS["product"]

    def run(x, fi, fo):
        out, delay, expected, _ = mi.simulate_product_resampler(x, fi, fo)
        return np.asarray(out)[:expected], delay

    for fi, fo in ((44_100, 48_000), (48_000, 44_100)):
        x = np.zeros(fi)
        x[fi // 2] = 1.0
        y, delay = run(x, fi, fo)
        assert delay == pins["delays"][f"{fi}->{fo}"]
        assert int(np.argmax(np.abs(y))) == pins["impulse_peak_index"][f"{fi}->{fo}"]
    noise = R.stopband_noise()
    y, _ = run(noise, 48_000, 44_100)
    swept = R.db_ratio(R.rms(R.steady(y, 44_100)), R.rms(R.steady(noise, 48_000)))
    assert abs(swept - pins["swept_noise_attenuation_db"]) <= 1e-10
    src = R.roundtrip_noise()
    up, _ = run(src, 44_100, 48_000)
    back, _ = run(up, 48_000, 44_100)
    n = min(src.size, back.size)
    err = back[4096 : n - 4096] - src[4096 : n - 4096]
    assert abs(R.db_ratio(R.rms(src[4096 : n - 4096]), R.rms(err)) - pins["roundtrip_snr_db"]) <= 1e-10
    assert abs(float(np.max(np.abs(err))) - pins["roundtrip_max_absolute_error"]) <= 1e-14
    y, _ = run(R.sine(44_100, 21_000.0, 2.0), 44_100, 48_000)
    image = R.db_ratio(R.tone_amplitude(y, 48_000, 44_100.0 - 21_000.0), R.tone_amplitude(y, 48_000, 21_000.0))
    assert abs(image - pins["worst_image_db"]) <= 1e-9


def test_long_stream_properties(mi):
    """Size-independent properties at a BASELINE-sized stream (10 s): linearity and shift-by-a-period.
    44.1 -> 48 kHz repeats its sub-sample phase every 147 input / 160 output frames."""
    from mic_eq_mi import mic_eq_core as core

    rng = np.random.default_rng(7)
    n = 441_000
    a = rng.standard_normal(n) * 0.1
    b = rng.standard_normal(n) * 0.1
    shifted = np.concatenate([np.zeros(147 * 64), a])[:n]
    batch = np.stack([a, b, a + b, shifted])
    out, delay, expected, blocks, _ = core.simulate_product_resampler_batch(batch, 44_100, 48_000)
    assert expected == 480_000 and out.shape[1] >= expected + delay
    assert np.max(np.abs(out[2] - (out[0] + out[1]))) <= 1e-13
    # the shifted copy is the original delayed by 160*64 output frames (positions accumulate rounding: ~1e-9)
    k = 160 * 64
    tail = 256  # the shifted copy is cut at n: its last sinc_len/2 * ratio frames see zeros instead of the end of `a`
    assert np.max(np.abs(out[3][k : expected - tail] - out[0][: expected - k - tail])) <= 1e-8
