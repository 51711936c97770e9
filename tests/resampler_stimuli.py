"""Deterministic stimuli and measurements of the reference's resampler evaluation
(python/tools/evaluate_resampler_quality.py:28-39, 115-150, 216-233, 326-344), restated for the tests:
the published numbers in tests/golden/resampler_report_pins.json are functions of exactly these signals."""
import math

import numpy as np

PASSBAND_HZ = (50.0, 100.0, 1_000.0, 5_000.0, 10_000.0, 15_000.0, 18_000.0, 20_000.0)
STOPBAND_HZ = (22_500.0, 23_000.0, 23_500.0)
IMAGE_TONES_HZ = (20_500.0, 21_000.0)


def rms(v) -> float:
    return float(np.sqrt(np.mean(np.square(v, dtype=np.float64))))


def db_ratio(num: float, den: float) -> float:
    return -300.0 if num <= 0.0 else 20.0 * math.log10(num / max(den, 1e-15))


def sine(fs: int, hz: float, seconds: float) -> np.ndarray:
    frames = int(round(fs * seconds))
    return 0.5 * np.sin(2.0 * np.pi * hz * (np.arange(frames, dtype=np.float64) / fs))


def steady(v: np.ndarray, fs: int) -> np.ndarray:
    margin = min(int(round(0.25 * fs)), max(0, v.size // 4))
    return v if margin == 0 else v[margin:-margin]


def shaped_noise(fs: int, low: float, high: float, seconds: float, seed: int, pink: bool) -> np.ndarray:
    frames = int(round(fs * seconds))
    freqs = np.fft.rfftfreq(frames, d=1.0 / fs)
    mask = (freqs >= low) & (freqs <= high)
    rng = np.random.default_rng(seed)
    spectrum = np.zeros(freqs.size, dtype=np.complex128)
    draw = rng.standard_normal(mask.sum()) + 1j * rng.standard_normal(mask.sum())
    spectrum[mask] = draw / np.sqrt(freqs[mask]) if pink else draw
    values = np.fft.irfft(spectrum, n=frames)
    return values * (0.2 / max(rms(values), 1e-15))


def stopband_noise() -> np.ndarray:
    return shaped_noise(48_000, 22_500.0, 23_900.0, 4.0, 0xA11A5, pink=False)


def roundtrip_noise() -> np.ndarray:
    return shaped_noise(44_100, 50.0, 20_000.0, 8.0, 0xA0D10, pink=True)


def tone_amplitude(v: np.ndarray, fs: int, hz: float) -> float:
    v = steady(v, fs)
    w = np.hanning(v.size)
    phase = np.exp(-2j * np.pi * hz * np.arange(v.size, dtype=np.float64) / fs)
    gain = float(np.sum(w)) / v.size
    return float(2.0 * np.abs(np.sum(v * w * phase)) / (v.size * max(gain, 1e-15)))
