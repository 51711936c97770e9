"""Deterministic synthetic inputs and waveform metrics shared by the CPU and GPU tests, smoke() and bench.py.

Nothing here is reference text and no reference file is read at run time.  Three kinds of content:

* the golden-KAT generator of `rust-core/src/audio/processor/tests.rs:1824-1851`, vectorised, and its per-stream variation
   -- this repository's own numpy formulation of the Rust test's arithmetic;
* DATA captured from the reference's evaluators by `tools/gen_golden.py` in the build container (tests/golden/*.json): the
  chain settings dicts and band lists they hand to `simulate_auto_eq_chain`, fingerprints of their stimuli, and the values
  their metric functions return on the oracle's output;
* this repository's own generators for those stimuli (held to the fingerprints by tests/test_golden_fixtures.py) and its
  own vectorised forms of the waveform metrics the reports define (held to the captured values by the same test).
"""
from __future__ import annotations

import copy
import json
import pathlib

import numpy as np

SAMPLE_RATE = 48_000
KAT_NOISE_STATE = 0x6A09E667F3BCC909
MASK64 = (1 << 64) - 1

_GOLDEN = pathlib.Path(__file__).resolve().parent / "golden"
LIMITER_FIXTURE = json.loads((_GOLDEN / "limiter_lookahead.json").read_text())
ALIASING_FIXTURE = json.loads((_GOLDEN / "dynamics_aliasing.json").read_text())


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


# ------------------------------------------------------------------------------------------------------------------
# Chain configurations of the reference's evaluators: data (tests/golden/*.json), captured at their call into the operator.
def limiter_settings(lookahead_ms: float) -> dict:
    """The settings dict `evaluation/limiter-lookahead-report.json` was rendered with, for one of its lookaheads; other
    lookaheads take the 2 ms dict with `limiter_lookahead_ms` replaced (the evaluator's only per-lookahead key)."""
    table = LIMITER_FIXTURE["settings"]
    key = f"{float(lookahead_ms):g}"
    settings = copy.deepcopy(table[key] if key in table else table["2"])
    settings["limiter_lookahead_ms"] = float(lookahead_ms)
    return settings


LIMITER_BANDS = [tuple(band) for band in LIMITER_FIXTURE["bands"]]
ALIASING_CASES = tuple((name, carrier, modulation) for name, carrier, modulation in ALIASING_FIXTURE["cases"])
ALIASING_SETTINGS = dict(ALIASING_FIXTURE["settings"])
ALIASING_BANDS = [tuple(band) for band in ALIASING_FIXTURE["bands"]]

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


# ------------------------------------------------------------------------------------------------------------------
# Stimuli of the two evaluators, regenerated from their descriptions (the reports' `configuration`
# blocks).  tests/test_golden_fixtures.py holds every one of them to the SHA-256 / head / tail / checkpoint fingerprints
# that tools/gen_golden.py took from the reference's own generators.
def _seconds(n: int, fs: int) -> np.ndarray:
    return np.arange(n) / fs


def _hann_burst_train(n: int, fs: int) -> np.ndarray:
    """25 ms Hann-windowed 6.5 kHz bursts at 1.35 x full scale, one every 190 ms from 0.25 s on (19 of them in 4 s)."""
    starts = ((0.25 + 0.19 * np.arange(19)) * fs).astype(np.int64)
    length = int(0.025 * fs)
    index = starts[:, None] + np.arange(length)[None, :]          # [burst][sample]: the bursts do not overlap
    carrier = np.sin(2.0 * np.pi * 6_500.0 * (index / fs))
    train = np.zeros(n, dtype=np.float64)
    train[index.ravel()] += ((1.35 * np.hanning(length))[None, :] * carrier).ravel()  # (added to silence: a window edge is +0.0, never -0.0)
    return train


def _impulse_train(n: int) -> np.ndarray:
    """+1.45 on every 997th sample, -1.35 on samples 499 mod 1553 (the negative train wins where they coincide)."""
    position = np.arange(n)
    return np.where(position % 1553 == 499, -1.35, np.where(position % 997 == 0, 1.45, 0.0))


def _clipped_two_tone_voice(n: int, fs: int) -> np.ndarray:
    """180 Hz + 2.3 kHz under a 2.1 Hz squared-sine swell, driven 25 % past full scale and hard-clipped."""
    t = _seconds(n, fs)
    tones = 0.72 * np.sin(2.0 * np.pi * 180.0 * t) + 0.46 * np.sin(2.0 * np.pi * 2_300.0 * t)
    swell = 0.45 + 0.55 * np.sin(2.0 * np.pi * 2.1 * t) ** 2
    return np.clip(tones * swell * 1.25, -1.0, 1.0)


def limiter_cases() -> dict[str, np.ndarray]:
    """The three controlled 4 s stimuli of `evaluation/limiter-lookahead-report.json` (float32)."""
    n = SAMPLE_RATE * 4
    made = {
        "controlled-sine-bursts": _hann_burst_train(n, SAMPLE_RATE),
        "controlled-impulses": _impulse_train(n),
        "controlled-clipped-voice": _clipped_two_tone_voice(n, SAMPLE_RATE),
    }
    return {name: made[name].astype(np.float32) for name in LIMITER_FIXTURE["stimuli"]}  # (the fixture's case order)


def aliasing_signal(sample_rate: int, carrier_hz: float, modulation_hz: float) -> np.ndarray:
    """One amplitude-modulated carrier of `evaluation/dynamics-aliasing-report.json`, 4 s at `sample_rate`: a squared-sine
    swell at `modulation_hz` plus a 1.5 ms exponential click every 173 ms, capped at 0.95."""
    n = int(4.0 * sample_rate)
    t = np.arange(n, dtype=np.float64) / sample_rate
    swell = 0.08 + 0.72 * np.square(0.5 + 0.5 * np.sin(2.0 * np.pi * modulation_hz * t))
    period = max(1, int(round(0.173 * sample_rate)))
    one_click = np.exp(-np.arange(period) / max(1.0, 0.0015 * sample_rate))  # the decay depends on the phase only: one period, repeated
    clicks = np.resize(one_click, n)
    level = np.clip(swell + 0.35 * clicks, 0.0, 0.95)
    return (level * np.sin(2.0 * np.pi * carrier_hz * t)).astype(np.float32)


def fingerprint_mismatch(x: np.ndarray, want: dict) -> str | None:
    """None when `x` is the stimulus `want` (a fingerprint written by tools/gen_golden.py) describes, else what differs."""
    import hashlib

    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.size != want["n"]:
        return f"length {x.size} != {want['n']}"
    if not np.array_equal(x[:64], np.asarray(want["head"], dtype=np.float32)):
        return "first 64 samples differ"
    if not np.array_equal(x[-64:], np.asarray(want["tail"], dtype=np.float32)):
        return "last 64 samples differ"
    for index, value in want["checkpoints"]:
        if x[index] != np.float32(value):
            return f"sample {index}: {x[index]!r} != {value!r}"
    if hashlib.sha256(x.astype("<f4").tobytes()).hexdigest() != want["sha256_f32le"]:
        return "SHA-256 over the float32 bytes differs"
    return None


# ------------------------------------------------------------------------------------------------------------------
# Waveform metrics of the limiter-lookahead report (`evaluation/limiter-lookahead-report.json`, "evaluation_contract"):
# this repository's vectorised forms.  tests/test_golden_fixtures.py checks them against the values the reference's own
# functions returned on the oracle's output (tests/golden/limiter_lookahead.json, "rows").
def _framed(x: np.ndarray, frame: int, hop: int) -> np.ndarray:
    return np.lib.stride_tricks.sliding_window_view(np.asarray(x, dtype=np.float64), frame)[::hop]


def gain_envelope_variation_db(reference: np.ndarray, aligned: np.ndarray) -> float:
    """Standard deviation (about the median) of the short-term gain output / input in dB: 2 ms frames at half overlap,
    frames whose input is below -40 dBFS left out."""
    frame = int(round(0.002 * SAMPLE_RATE))
    if reference.size < frame:
        return 0.0
    rms_in = np.sqrt(np.mean(np.square(_framed(reference, frame, frame // 2)), axis=1))
    rms_out = np.sqrt(np.mean(np.square(_framed(aligned[: reference.size], frame, frame // 2)), axis=1))
    audible = rms_in >= 10.0 ** (-40.0 / 20.0)
    if not audible.any():
        return 0.0
    gain_db = 20.0 * np.log10(np.maximum(rms_out[audible], 1e-12) / np.maximum(rms_in[audible], 1e-12))
    return float(np.std(gain_db - np.median(gain_db)))


def transient_indices(audio: np.ndarray, limit: int = 16) -> np.ndarray:
    """The `limit` steepest sample-to-sample steps that lie at least 6 ms apart, steepest first (ties: earliest), sorted."""
    audio = np.asarray(audio, dtype=np.float64)
    step = np.abs(np.diff(audio, prepend=0.0))
    guard = int(round(0.006 * SAMPLE_RATE))
    taken = np.zeros(audio.size, dtype=bool)   # samples closer than `guard` to a transient already picked
    picked = []
    for index in np.argsort(-step, kind="stable"):
        if taken[index]:
            continue
        picked.append(int(index))
        if len(picked) == limit:
            break
        taken[max(0, index - guard + 1) : index + guard] = True
    return np.asarray(sorted(picked), dtype=np.int64)


def transient_error_db(reference: np.ndarray, aligned: np.ndarray, indices: np.ndarray) -> float:
    """Median over the transients of the shape error in a +-4 ms window: the output minus its least-squares projection on
    the input, relative to the output, in dB (windows shorter than 8 samples are skipped; -240 dB when none is left)."""
    reference = np.asarray(reference, dtype=np.float64)
    aligned = np.asarray(aligned, dtype=np.float64)
    radius = int(round(0.004 * SAMPLE_RATE))
    floor = 1e-12
    shape_db = []
    for centre in np.asarray(indices, dtype=np.int64):
        lo, hi = max(0, int(centre) - radius), min(reference.size, int(centre) + radius + 1)
        if hi - lo < 8:
            continue
        x, y = reference[lo:hi], aligned[lo:hi]
        residual = y - (float(np.dot(x, y)) / max(float(np.dot(x, x)), floor)) * x
        rms = [float(np.sqrt(np.mean(np.square(v)))) for v in (residual, y)]
        shape_db.append(20.0 * np.log10(max(rms[0], floor) / max(rms[1], floor)))
    return float(np.median(shape_db)) if shape_db else -240.0


# Metrics of the dynamics-aliasing report (48 kHz render against a 192 kHz render brought down to 48 kHz).
def align_renders(reference: np.ndarray, candidate: np.ndarray, max_lag: int = 256) -> tuple[np.ndarray, np.ndarray, int]:
    """Shift `candidate` against `reference` by the lag (|lag| <= max_lag) of largest absolute cross-correlation over the
    first second, and cut both to their overlap."""
    probe = min(reference.size, candidate.size, SAMPLE_RATE)
    r = reference[:probe] - np.mean(reference[:probe])
    c = candidate[:probe] - np.mean(candidate[:probe])
    spectrum = np.fft.rfft(c, 2 * probe) * np.conj(np.fft.rfft(r, 2 * probe))
    xcorr = np.fft.irfft(spectrum, 2 * probe)  # xcorr[k] = sum_n c[n + k] r[n]; negative lags wrap to the end
    lags = np.arange(-max_lag, max_lag + 1)
    lag = int(lags[int(np.argmax(np.abs(xcorr[lags % (2 * probe)])))])
    if lag >= 0:
        count = min(reference.size, candidate.size - lag)
        return reference[:count], candidate[lag : lag + count], lag
    count = min(reference.size + lag, candidate.size)
    return reference[-lag : -lag + count], candidate[:count], lag


def relative_error_db(reference: np.ndarray, candidate: np.ndarray) -> float:
    rms = [float(np.sqrt(np.mean(np.square(v)))) for v in (candidate - reference, reference)]
    return float(20.0 * np.log10(max(rms[0], 1e-12) / max(rms[1], 1e-12)))


def folded_error_db(reference: np.ndarray, candidate: np.ndarray, carrier_hz: float, modulation_hz: float) -> float:
    """Energy of the (Hann-windowed) difference outside the sidebands carrier +- k x modulation, k <= 12 -- where only folded
    products can land -- relative to the reference's energy, in dB."""
    window = np.hanning(reference.size)
    frequency = np.fft.rfftfreq(reference.size, 1.0 / SAMPLE_RATE)
    centres = carrier_hz + modulation_hz * np.arange(-12, 13)
    centres = centres[(centres >= 0.0) & (centres <= SAMPLE_RATE / 2)]
    half_width = max(8.0, 0.12 * modulation_hz)
    expected = (np.abs(frequency[:, None] - centres[None, :]) <= half_width).any(axis=1)
    error_power = np.square(np.abs(np.fft.rfft((candidate - reference) * window)))
    reference_power = np.square(np.abs(np.fft.rfft(reference * window)))
    return float(10.0 * np.log10(max(float(error_power[~expected].sum()), 1e-24) / max(float(reference_power.sum()), 1e-24)))


def aliasing_case_metrics(base_output: np.ndarray, reference_output: np.ndarray, carrier_hz: float, modulation_hz: float) -> dict:
    """The waveform rows of one aliasing case from its two renders (48 kHz; 192 kHz), as the report defines them."""
    from scipy.signal import resample_poly

    down = resample_poly(np.asarray(reference_output, dtype=np.float64), 1, 4)
    reference, candidate, lag = align_renders(down, np.asarray(base_output, dtype=np.float64))
    trim = SAMPLE_RATE // 2
    if reference.size > 2 * trim:
        reference, candidate = reference[trim:-trim], candidate[trim:-trim]
    return {"alignment_lag_samples": lag, "relative_waveform_error_db": relative_error_db(reference, candidate),
            "folded_out_of_expected_error_db": folded_error_db(reference, candidate, carrier_hz, modulation_hz)}
