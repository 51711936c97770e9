"""Pin the resampler restatement (oracle/af_resampler.c) against the reference's published measurements.

rubato's source is not available, so there are no sample vectors; but evaluation/resampler-quality-report.json
holds 16-digit measurements of deterministic stimuli through `simulate_product_resampler`.  A restatement
that differs anywhere (window, cutoff, sinc table normalisation, sub-sample positions, cubic interpolation,
chunking/flush) cannot reproduce them: they agree here to 1e-12 relative.
"""
import json
import pathlib

import numpy as np
import pytest

import resampler_stimuli as R

PINS = json.loads((pathlib.Path(__file__).parent / "golden" / "resampler_report_pins.json").read_text())


def _run(o, x, fi, fo, sinc_len=None, window=None):
    y, delay, expected, blocks = o.simulate_product_resampler(x, fi, fo, 1024, sinc_len, window)
    assert y.size >= expected
    return y[:expected], delay, blocks


def _close(got, want, rel=1e-11):
    assert abs(got - want) <= rel * max(1.0, abs(want)), (got, want)


def test_identified_cutoffs(oracle):
    assert np.float32(oracle.resampler_calculate_cutoff(128, "blackman")).view(np.uint32) == 0x3F73E7B4
    assert np.float32(oracle.resampler_calculate_cutoff(128, "blackman_harris_squared")).view(np.uint32) == 0x3F650CE0
    assert np.float32(oracle.resampler_calculate_cutoff(256, "blackman_harris_squared")).view(np.uint32) == 0x3F72722D


def test_delay_expected_frames_blocks_and_impulse(oracle):
    """tests.rs:194-207 (expected = 48000 for 44100 frames, output >= delay + expected) and the report's
    impulse / long-stream rows."""
    pins = PINS["product"]
    for fi, fo in ((44_100, 48_000), (48_000, 44_100)):
        key = f"{fi}->{fo}"
        x = np.zeros(fi)
        x[fi // 2] = 1.0
        y, delay, expected, _ = oracle.simulate_product_resampler(x, fi, fo)
        assert expected == fo and delay == pins["delays"][key] and y.size >= delay + expected
        assert int(np.argmax(np.abs(y[:expected]))) == pins["impulse_peak_index"][key]
        y, delay, expected, blocks = oracle.simulate_product_resampler(np.zeros(fi * 60), fi, fo)
        assert blocks == pins["long_stream_blocks"][key] and expected == fo * 60 and y.size >= expected
        assert not np.any(y)


def test_stopband_and_image_rejection(oracle):
    pins = PINS["product"]
    noise = R.stopband_noise()
    y, _, _ = _run(oracle, noise, 48_000, 44_100)
    swept = R.db_ratio(R.rms(R.steady(y, 44_100)), R.rms(R.steady(noise, 48_000)))
    _close(swept, pins["swept_noise_attenuation_db"])
    tones = []
    for hz in R.STOPBAND_HZ:
        s = R.sine(48_000, hz, 2.0)
        y, _, _ = _run(oracle, s, 48_000, 44_100)
        tones.append(R.db_ratio(R.rms(R.steady(y, 44_100)), R.rms(R.steady(s, 48_000))))
    _close(max(swept, *tones), pins["worst_alias_db"])
    images = []
    for hz in R.IMAGE_TONES_HZ:
        y, _, _ = _run(oracle, R.sine(44_100, hz, 2.0), 44_100, 48_000)
        images.append(R.db_ratio(R.tone_amplitude(y, 48_000, 44_100.0 - hz), R.tone_amplitude(y, 48_000, hz)))
    _close(max(images), pins["worst_image_db"])


@pytest.mark.parametrize("name", ["product", "legacy-blackman-harris-squared-128", "high-rejection-blackman-harris-squared-256"])
def test_roundtrip_and_passband(oracle, name):
    pins = PINS[name]
    cfg = (None, None) if name == "product" else (pins["sinc_len"], pins["window"])
    src = R.roundtrip_noise()
    up, d_up, _ = _run(oracle, src, 44_100, 48_000, *cfg)
    back, d_down, _ = _run(oracle, up, 48_000, 44_100, *cfg)
    if "delays" in pins:
        assert (d_up, d_down) == (pins["delays"]["44100->48000"], pins["delays"]["48000->44100"])
    if "roundtrip_frames" in pins:
        assert [src.size, up.size, back.size] == pins["roundtrip_frames"]
    n = min(src.size, back.size)
    err = back[4096 : n - 4096] - src[4096 : n - 4096]
    _close(R.db_ratio(R.rms(src[4096 : n - 4096]), R.rms(err)), pins["roundtrip_snr_db"])
    _close(float(np.max(np.abs(err))), pins["roundtrip_max_absolute_error"])
    for fi, fo in ((44_100, 48_000), (48_000, 44_100)):
        gains = []
        for hz in R.PASSBAND_HZ:
            s = R.sine(fi, hz, 1.5)
            y, _, _ = _run(oracle, s, fi, fo, *cfg)
            gains.append(R.db_ratio(R.rms(R.steady(y, fo)), R.rms(R.steady(s, fi))))
        key = f"{fi}->{fo}"
        _close(max(abs(g) for g in gains), pins["passband_max_absolute_error_db"][key], rel=1e-9)
        if "passband_ripple_db" in pins:
            _close(max(gains) - min(gains), pins["passband_ripple_db"][key], rel=1e-9)
