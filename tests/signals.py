"""Deterministic synthetic inputs shared by the CPU and GPU tests and bench.py.

Each generator restates the formula of a reference fixture (cited) in numpy; no
reference file is read at run time.
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 48_000
KAT_NOISE_STATE = 0x6A09E667F3BCC909
MASK64 = (1 << 64) - 1


def kat_signal(n_blocks: int, noise_state: int = KAT_NOISE_STATE, f0: float = 180.0, phrase_hz: float = 1.7) -> np.ndarray:
    """Golden-KAT input (rust-core/src/audio/processor/tests.rs:1824-1851), vectorised.

    f0 / phrase_hz / noise_state generalise it per stream; the
    defaults are the KAT itself.  Tones sit at f0, 2*f0 and 15*f0 (180/360/2700 Hz).
    """
    n = n_blocks * 480
    idx = np.arange(n, dtype=np.float64)
    time = idx / 48000.0
    # LCG: state_{k} = a^k s0 + c (a^k-1)/(a-1)  -- iterate in uint64 blocks for speed
    a = np.uint64(6364136223846793005)
    c = np.uint64(1442695040888963407)
    states = np.empty(n, dtype=np.uint64)
    s = np.uint64(noise_state & MASK64)
    with np.errstate(over="ignore"):
        # jump table: process sequentially in chunks using vectorised affine powers
        chunk = 4096
        mul = np.empty(chunk, dtype=np.uint64)
        add = np.empty(chunk, dtype=np.uint64)
        m, d = np.uint64(1), np.uint64(0)
        for k in range(chunk):
            m = m * a
            d = d * a + c
            mul[k] = m
            add[k] = d
        for start in range(0, n, chunk):
            ln = min(chunk, n - start)
            states[start : start + ln] = mul[:ln] * s + add[:ln]
            s = states[start + ln - 1]
    noise = ((states >> np.uint64(40)).astype(np.uint32).astype(np.float64) / float((1 << 24) - 1) * 2.0 - 1.0) * 0.012
    phrase = 0.25 + 0.75 * np.abs(np.sin(2.0 * np.pi * phrase_hz * time))
    block_index = np.arange(n) // 480
    gate = (((block_index // 12) % 5) == 2).astype(np.float64)
    voiced = (
        0.30 * np.sin(2.0 * np.pi * f0 * time)
        + 0.14 * np.sin(2.0 * np.pi * (2.0 * f0) * time)
        + 0.08 * np.sin(2.0 * np.pi * (15.0 * f0) * time)
    )
    x = phrase * voiced + gate * 0.35 * np.sin(2.0 * np.pi * 7200.0 * time) + noise
    return x.astype(np.float32)


def stream_params(i: int) -> tuple[int, float, float]:
    """Per-stream variation of the KAT generator."""
    state = KAT_NOISE_STATE ^ ((i * 0x9E3779B97F4A7C15) & MASK64)
    f0 = 180.0 * 2.0 ** (((i % 25) - 12) / 24.0)
    phrase = 1.7 * (1.0 + 0.01 * (i % 7))
    return state, f0, phrase


def batch_signal(n_streams: int, n_blocks: int) -> np.ndarray:
    """S3 batch: [n_streams, n_blocks*480] f32, stream-major."""
    out = np.empty((n_streams, n_blocks * 480), dtype=np.float32)
    for i in range(n_streams):
        st, f0, ph = stream_params(i)
        out[i] = kat_signal(n_blocks, st, f0, ph)
    return out


def limiter_cases() -> dict[str, np.ndarray]:
    """The three controlled fixtures of python/tools/evaluate_limiter_lookahead.py:34-61."""
    sample_count = SAMPLE_RATE * 4
    time = np.arange(sample_count) / SAMPLE_RATE
    sine_bursts = np.zeros(sample_count, dtype=np.float64)
    for start_s in np.arange(0.25, 3.75, 0.19):
        start = int(start_s * SAMPLE_RATE)
        length = int(0.025 * SAMPLE_RATE)
        envelope = np.hanning(length)
        sine_bursts[start : start + length] += 1.35 * envelope * np.sin(2.0 * np.pi * 6_500.0 * time[start : start + length])
    impulses = np.zeros(sample_count, dtype=np.float64)
    impulses[::997] = 1.45
    impulses[499::1553] = -1.35
    clipped_voice = 0.72 * np.sin(2.0 * np.pi * 180.0 * time) + 0.46 * np.sin(2.0 * np.pi * 2_300.0 * time)
    clipped_voice *= 0.45 + 0.55 * np.sin(2.0 * np.pi * 2.1 * time) ** 2
    clipped_voice = np.clip(clipped_voice * 1.25, -1.0, 1.0)
    return {
        "controlled-sine-bursts": np.asarray(sine_bursts, dtype=np.float32),
        "controlled-impulses": np.asarray(impulses, dtype=np.float32),
        "controlled-clipped-voice": np.asarray(clipped_voice, dtype=np.float32),
    }


def limiter_settings(lookahead_ms: float) -> dict:
    """python/tools/evaluate_limiter_lookahead.py:143-161."""
    return {
        "deesser_enabled": False,
        "compressor_enabled": True,
        "compressor_threshold_db": -20.0,
        "compressor_ratio": 4.0,
        "compressor_attack_ms": 10.0,
        "compressor_release_ms": 200.0,
        "compressor_makeup_gain_db": 0.0,
        "compressor_adaptive_release": False,
        "compressor_auto_makeup_enabled": False,
        "compressor_sidechain_highpass_enabled": True,
        "limiter_enabled": True,
        "limiter_ceiling_db": -0.5,
        "limiter_release_ms": 50.0,
        "limiter_careful_output_enabled": True,
        "limiter_lookahead_ms": lookahead_ms,
        "return_output_audio": True,
    }


LIMITER_BANDS = [(80.0 * 1.75**index, 0.0, 1.0) for index in range(10)]


def aliasing_signal(sample_rate: int, carrier_hz: float, modulation_hz: float) -> np.ndarray:
    """python/tools/evaluate_dynamics_aliasing.py:30-43."""
    duration = 4.0
    time = np.arange(int(duration * sample_rate), dtype=np.float64) / sample_rate
    slow_envelope = 0.08 + 0.72 * np.square(0.5 + 0.5 * np.sin(2.0 * np.pi * modulation_hz * time))
    transient_period = max(1, int(round(0.173 * sample_rate)))
    transient_phase = np.arange(time.size) % transient_period
    transient = np.exp(-transient_phase / max(1.0, 0.0015 * sample_rate))
    envelope = np.clip(slow_envelope + 0.35 * transient, 0.0, 0.95)
    return np.asarray(envelope * np.sin(2.0 * np.pi * carrier_hz * time), dtype=np.float32)


ALIASING_CASES = (
    ("carrier_8k", 8_000.0, 37.0),
    ("carrier_11k", 11_000.0, 73.0),
    ("carrier_15k", 15_000.0, 113.0),
    ("carrier_18k", 18_000.0, 157.0),
)
ALIASING_SETTINGS = {
    "deesser_enabled": False,
    "compressor_enabled": True,
    "compressor_threshold_db": -24.0,
    "compressor_ratio": 8.0,
    "compressor_attack_ms": 0.5,
    "compressor_release_ms": 50.0,
    "compressor_makeup_gain_db": 0.0,
    "compressor_adaptive_release": False,
    "compressor_sidechain_highpass_enabled": False,
    "limiter_enabled": False,
    "return_output_audio": True,
}
ALIASING_BANDS = [(100.0 * 1.7**index, 0.0, 1.0) for index in range(10)]

DEFAULT_TYPED_BANDS = [
    ("low_shelf", 80.0, 0.0, 1.41, 12, True),
    ("bell", 160.0, 0.0, 1.41, 12, True),
    ("bell", 320.0, 0.0, 1.41, 12, True),
    ("bell", 640.0, 0.0, 1.41, 12, True),
    ("bell", 1280.0, 0.0, 1.41, 12, True),
    ("bell", 2500.0, 0.0, 1.41, 12, True),
    ("bell", 5000.0, 0.0, 1.41, 12, True),
    ("bell", 8000.0, 0.0, 1.41, 12, True),
    ("bell", 12000.0, 0.0, 1.41, 12, True),
    ("high_shelf", 16000.0, 0.0, 1.41, 12, True),
]


# ---- metric code of python/tools/evaluate_limiter_lookahead.py:214-283 (restated) ----
def gain_envelope_variation_db(reference: np.ndarray, aligned: np.ndarray) -> float:
    import math

    frame_samples = int(round(0.002 * SAMPLE_RATE))
    hop_samples = frame_samples // 2
    gains_db = []
    for start in range(0, reference.size - frame_samples + 1, hop_samples):
        ref = np.asarray(reference[start : start + frame_samples], dtype=np.float64)
        out = np.asarray(aligned[start : start + frame_samples], dtype=np.float64)
        reference_rms = float(np.sqrt(np.mean(np.square(ref))))
        if reference_rms < 10.0 ** (-40.0 / 20.0):
            continue
        output_rms = float(np.sqrt(np.mean(np.square(out))))
        gains_db.append(20.0 * math.log10(max(output_rms, 1e-12) / max(reference_rms, 1e-12)))
    if not gains_db:
        return 0.0
    values = np.asarray(gains_db, dtype=np.float64)
    return float(np.std(values - np.median(values)))


def transient_indices(audio: np.ndarray, limit: int = 16) -> np.ndarray:
    derivative = np.abs(np.diff(np.asarray(audio, dtype=np.float64), prepend=0.0))
    order = np.argsort(-derivative, kind="stable")
    separation = int(round(0.006 * SAMPLE_RATE))
    selected: list[int] = []
    for raw_index in order:
        index = int(raw_index)
        if all(abs(index - existing) >= separation for existing in selected):
            selected.append(index)
        if len(selected) == limit:
            break
    return np.asarray(sorted(selected), dtype=np.int64)


def transient_error_db(reference: np.ndarray, aligned: np.ndarray, indices: np.ndarray) -> float:
    import math

    radius = int(round(0.004 * SAMPLE_RATE))
    errors = []
    for index in indices:
        start = max(0, int(index) - radius)
        end = min(reference.size, int(index) + radius + 1)
        ref = np.asarray(reference[start:end], dtype=np.float64)
        out = np.asarray(aligned[start:end], dtype=np.float64)
        if ref.size < 8:
            continue
        denominator = float(np.dot(ref, ref))
        scale = float(np.dot(ref, out) / max(denominator, 1e-12))
        error = out - scale * ref
        errors.append(
            20.0
            * math.log10(
                max(float(np.sqrt(np.mean(np.square(error)))), 1e-12)
                / max(float(np.sqrt(np.mean(np.square(out)))), 1e-12)
            )
        )
    return float(np.median(errors)) if errors else -240.0
