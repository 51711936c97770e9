"""Drop-in for the offline operator surface of ``mic_eq.mic_eq_core`` on the MI355X engine.

Function names, argument meaning, returned dict keys and error behaviour follow the PyO3
module of the reference (rust-core/src/lib.rs:99-350, audio/processor/python_api.rs:118-714,
typed in python/mic_eq/mic_eq_core.pyi:202-258).  The audio path is the HIP engine behind
``libaudioforge_mi.so``; this file only (a) replays the reference's setter sequence on an
engine and (b) folds the per-block rows the kernels emit into the dict statistics
(python_api.rs:578-713).  There is no CPU audio path here.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Any, Mapping, Sequence

import numpy as np

from . import _lib
from ._lib import BlockStats, EqBandConfig

NUM_BANDS = 10
EQ_TYPE_IDS = {"low_shelf": 0, "bell": 1, "high_shelf": 2, "notch": 3, "high_pass": 4, "low_pass": 5}
CAREFUL_OUTPUT_CEILING_DB = -1.5  # audio/processor/control.rs:772
STATS_DTYPE = np.dtype(
    [
        ("input_sample_peak", "<f4"),
        ("output_sample_peak", "<f4"),
        ("true_peak_limiter_input_peak", "<f4"),
        ("output_true_peak", "<f4"),
        ("limiter_peak_gain_reduction_db", "<f4"),
        ("true_peak_limiter_gain_reduction_db", "<f4"),
        ("compressor_gain_reduction_db", "<f4"),
        ("deesser_gain_reduction_db", "<f4"),
        ("input_square_sum", "<f8"),
        ("output_square_sum", "<f8"),
        ("true_peak_limited_events", "<u4"),
        ("non_finite_output", "<u4"),
        ("compressor_makeup_gain_db", "<f4"),
        ("auto_makeup_activity", "<f4"),
        ("auto_makeup_reliability", "<f4"),
        ("reserved", "<f4"),
    ]
)
assert STATS_DTYPE.itemsize == C.sizeof(BlockStats)

f32 = np.float32


# ------------------------------------------------------------------------------- engine
class Engine:
    """N independent reference chains (OfflineDspBlockProcessor) resident on one GPU.

    Setter methods are the C ABI entry points with the ``af_`` prefix dropped, e.g.
    ``engine.compressor_set_threshold(-20.0)`` or ``engine.eq_set_band_gain(2, 3.0)``.
    """

    def __init__(self, sample_rate: float = 48_000.0, n_streams: int = 1, device: int = 0):
        self._lib = _lib.load()
        handle = C.c_void_p()
        _lib.check(self._lib.af_engine_create(float(sample_rate), int(n_streams), int(device), C.byref(handle)))
        self._h = handle
        self._apply_variant_override()
        self.sample_rate = float(sample_rate)
        self.n_streams = int(n_streams)
        self.device = int(device)

    def _apply_variant_override(self) -> None:
        """AF_KERNEL_VARIANT=lane | quad | staged | roles | ring-<waves>x<chunk> pins the kernel (tests and tuning runs)."""
        import os

        variant = os.environ.get("AF_KERNEL_VARIANT", "")
        if variant == "lane":
            _lib.check(self._lib.af_engine_set_kernel(self._h, _lib.KERNEL_LANE_PER_STREAM))
        elif variant == "staged":
            _lib.check(self._lib.af_engine_set_kernel(self._h, _lib.KERNEL_STAGED))
        elif variant == "roles":
            _lib.check(self._lib.af_engine_set_kernel(self._h, _lib.KERNEL_ROLES))
        elif variant.startswith("quad"):  # quad | quad-<waves>
            _lib.check(self._lib.af_engine_set_kernel(self._h, _lib.KERNEL_QUAD))
            if "-" in variant:
                _lib.check(self._lib.af_engine_set_ring_variant(self._h, int(variant.split("-")[1]), 4))
        elif variant.startswith("ring-"):
            waves, chunk = (int(v) for v in variant[5:].split("x"))
            _lib.check(self._lib.af_engine_set_kernel(self._h, _lib.KERNEL_PHASED))
            _lib.check(self._lib.af_engine_set_ring_variant(self._h, waves, chunk))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.af_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getattr__(self, name: str):
        # engine.<x>(...) -> af_<x>(handle, ...) / af_engine_<x>(handle, ...)
        for symbol in (f"af_{name}", f"af_engine_{name}"):
            if symbol in _lib.SIGNATURES:
                fn = getattr(self._lib, symbol)
                is_status = symbol not in _lib.VALUE_FUNCTIONS

                def call(*args, _fn=fn, _is_status=is_status):
                    rc = _fn(self._h, *args)
                    if _is_status:
                        _lib.check(rc)
                        return None
                    return rc

                return call
        raise AttributeError(name)

    def eq_set_band_config_tuple(self, band: int, cfg: tuple) -> None:
        name, freq, gain, q, slope, enabled = cfg
        c = EqBandConfig(EQ_TYPE_IDS[name], float(freq), float(gain), float(q), int(slope), int(bool(enabled)))
        _lib.check(self._lib.af_eq_set_band_config(self._h, band, C.byref(c)))

    # -- processing ---------------------------------------------------------------
    def process(self, audio: np.ndarray, layout: int = _lib.LAYOUT_STREAM_MAJOR) -> np.ndarray:
        """Host arrays: [n_streams, n] (stream-major) or [n, n_streams] (time-major) float32."""
        a = np.ascontiguousarray(audio, dtype=np.float32)
        if a.ndim == 1:
            a = a.reshape(1, -1) if layout == _lib.LAYOUT_STREAM_MAJOR else a.reshape(-1, 1)
        n_streams, n = (a.shape if layout == _lib.LAYOUT_STREAM_MAJOR else a.shape[::-1])
        if n_streams != self.n_streams:
            raise ValueError(f"expected {self.n_streams} streams, got {n_streams}")
        fp = C.POINTER(C.c_float)
        if layout == _lib.LAYOUT_STREAM_MAJOR:
            # with the suppressor on a call returns the whole 480-sample frames that are complete (rnnoise.rs:114-164):
            # floor((pending + n) / 480) * 480 samples per stream, the rest waits in the engine
            stride = int(n) + 480
            out = np.empty((n_streams, stride), dtype=np.float32)
            n_out = C.c_int64(0)
            _lib.check(self._lib.af_engine_stream_host(self._h, a.ctypes.data_as(fp), int(n), out.ctypes.data_as(fp), stride,
                                                       C.byref(n_out)))
            return np.ascontiguousarray(out[:, : n_out.value])
        out = np.empty_like(a)
        _lib.check(self._lib.af_engine_process_host(self._h, a.ctypes.data_as(fp), out.ctypes.data_as(fp), int(n), layout))
        return out

    def suppressor_trace(self) -> np.ndarray:
        """[frames, n_streams, 2] int32: (silence flag, pitch index) of every frame of the last process call
        (needs ``suppressor_set_trace_enabled(1)`` before the call)."""
        frames = int(self._lib.af_suppressor_trace_frames(self._h))
        out = np.zeros((frames, self.n_streams, 2), dtype=np.int32)
        if frames:
            _lib.check(self._lib.af_suppressor_read_trace(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), frames))
        return out

    def process_device(self, in_ptr: int, out_ptr: int, n_samples: int, stream_stride: int,
                       layout: int = _lib.LAYOUT_STREAM_MAJOR, hip_stream: int = 0) -> None:
        """Device pointers (e.g. ``tensor.data_ptr()``); asynchronous on ``hip_stream``."""
        _lib.check(self._lib.af_engine_process_device(self._h, C.c_void_p(in_ptr), C.c_void_p(out_ptr), int(n_samples),
                                                      int(stream_stride), int(layout), C.c_void_p(hip_stream)))

    def block_stats(self) -> np.ndarray:
        """Structured array [blocks, n_streams] of the last process call."""
        blocks = int(self._lib.af_engine_last_block_count(self._h))
        rows = np.zeros((blocks, self.n_streams), dtype=STATS_DTYPE)
        if blocks:
            _lib.check(self._lib.af_engine_read_block_stats(self._h, rows.ctypes.data_as(C.POINTER(BlockStats)), rows.size))
        return rows

    def last_kernel_ms(self) -> tuple[float, int]:
        ms, launches = C.c_double(0.0), C.c_int32(0)
        _lib.check(self._lib.af_engine_last_kernel_ms(self._h, C.byref(ms), C.byref(launches)))
        return ms.value, launches.value

    def last_chain_launch_ms(self) -> tuple[float, float, int]:
        """(summed ms of the chain launches of the last call, 0.0, number of launches)."""
        first, tail, segments = C.c_double(0.0), C.c_double(0.0), C.c_int32(0)
        _lib.check(self._lib.af_engine_last_chain_launch_ms(self._h, C.byref(first), C.byref(tail), C.byref(segments)))
        return first.value, tail.value, segments.value


# ---------------------------------------------------------------------- argument checks
def _audio_1d(audio) -> np.ndarray:
    if not isinstance(audio, np.ndarray) or audio.dtype != np.float32 or audio.ndim != 1:
        raise TypeError("audio must be a 1-D numpy.float32 array")
    if not audio.flags.c_contiguous:
        raise ValueError("audio must be a contiguous float32 array")
    return audio


def _bands_v2(bands, sample_rate: float):
    """parse_eq_v2_bands, lib.rs:154-189."""
    if not np.isfinite(sample_rate) or sample_rate <= 0.0:
        raise ValueError("sample_rate must be finite and positive")
    if len(bands) != NUM_BANDS:
        raise ValueError(f"expected {NUM_BANDS} EQ bands, got {len(bands)}")
    arr = (EqBandConfig * NUM_BANDS)()
    lib = _lib.load()
    for index, (name, freq, gain, q, slope, enabled) in enumerate(bands):
        if name not in EQ_TYPE_IDS:
            raise ValueError(f"band {index} has unsupported EQ filter type: {name}")
        arr[index] = EqBandConfig(EQ_TYPE_IDS[name], float(freq), float(gain), float(q), int(slope), int(bool(enabled)))
        _lib.check(lib.af_eq_band_config_validate(C.byref(arr[index]), index, float(sample_rate)))
    return arr


def _get(settings: Mapping[str, object] | None, key: str, default):
    if settings is not None and key in settings and settings[key] is not None:
        value = settings[key]
        if isinstance(default, bool):
            if not isinstance(value, (bool, np.bool_)):
                raise TypeError(f"settings[{key!r}] must be a bool")
            return bool(value)
        return float(value)
    return default


# ---------------------------------------------------- python_api.rs:54-111 statistics
def _linear_to_db(value) -> np.float32:
    return f32(20.0) * np.log10(np.maximum(f32(value), f32(1.0e-12)), dtype=np.float32)


def _percentile(values: np.ndarray, percentile: float) -> np.float32:
    values = np.sort(np.asarray(values, dtype=np.float32))
    if values.size == 0:
        return f32(0.0)
    position = f32(values.size - 1) * f32(min(max(percentile, 0.0), 1.0))
    lower = int(np.floor(position))
    upper = int(np.ceil(position))
    if lower == upper:
        return values[lower]
    fraction = position - f32(lower)
    return values[lower] + fraction * (values[upper] - values[lower])


def _pumping_score(trace: np.ndarray, cadence_hz: float) -> np.float32:
    trace = np.asarray(trace, dtype=np.float32)
    if trace.size < 3 or not np.isfinite(cadence_hz) or cadence_hz <= 0.0:
        return f32(0.0)
    pi = f32(np.pi)
    dt = f32(1.0) / f32(cadence_hz)
    highpass_rc = f32(1.0) / (f32(2.0) * pi * f32(2.0))
    lowpass_rc = f32(1.0) / (f32(2.0) * pi * f32(8.0))
    highpass_alpha = highpass_rc / (highpass_rc + dt)
    lowpass_alpha = dt / (lowpass_rc + dt)
    previous = trace[0]
    highpass = f32(0.0)
    bandpass = f32(0.0)
    bandpass_abs = np.empty(trace.size - 1, dtype=np.float32)
    deltas = np.empty(trace.size - 1, dtype=np.float32)
    for i in range(1, trace.size):
        value = trace[i]
        if not np.isfinite(value):
            return f32(np.inf)
        highpass = highpass_alpha * (highpass + value - previous)
        bandpass = bandpass + lowpass_alpha * (highpass - bandpass)
        bandpass_abs[i - 1] = abs(bandpass)
        deltas[i - 1] = abs(value - previous)
        previous = value
    robust_limit = _percentile(bandpass_abs, 0.95)
    total = f32(0.0)
    for v in np.minimum(bandpass_abs, robust_limit):
        total = total + v * v
    robust_rms = np.sqrt(total / f32(bandpass_abs.size))
    return f32(robust_rms + _percentile(deltas, 0.95))


def chain_diagnostics(rows: np.ndarray, block_lengths: np.ndarray, effective_ceiling_db: float) -> dict[str, Any]:
    """Fold one stream's per-block rows into the dict of python_api.rs:578-713."""
    n_rows = rows.shape[0]
    in_sq_total = 0.0
    out_sq_total = 0.0
    for k in range(n_rows):  # sequential f64 accumulation order of python_api.rs:519-571
        in_sq_total += float(rows["input_square_sum"][k])
        out_sq_total += float(rows["output_square_sum"][k])
    samples = int(block_lengths.sum())
    input_rms = f32(np.sqrt(in_sq_total / samples)) if samples > 0 else f32(0.0)
    output_rms = f32(np.sqrt(out_sq_total / samples)) if samples > 0 else f32(0.0)
    lengths = block_lengths.astype(np.float64)
    block_in_rms = np.sqrt(rows["input_square_sum"] / lengths).astype(np.float32)
    block_out_rms = np.sqrt(rows["output_square_sum"] / lengths).astype(np.float32)
    in_db = _linear_to_db(block_in_rms)
    out_db = _linear_to_db(block_out_rms)
    comp = rows["compressor_gain_reduction_db"].astype(np.float32)
    dees = rows["deesser_gain_reduction_db"].astype(np.float32)

    def fmax(field):
        return f32(rows[field].max()) if n_rows else f32(0.0)

    input_sample_peak = max(f32(0.0), fmax("input_sample_peak"))
    output_sample_peak = max(f32(0.0), fmax("output_sample_peak"))
    pre_limiter_true_peak = max(f32(0.0), fmax("true_peak_limiter_input_peak"))
    output_true_peak = max(f32(0.0), fmax("output_true_peak"))
    output_sample_peak_db = _linear_to_db(output_sample_peak)
    pre_limiter_true_peak_db = _linear_to_db(pre_limiter_true_peak)
    output_true_peak_db = _linear_to_db(output_true_peak)
    ceiling = f32(effective_ceiling_db)

    input_floor_db = _percentile(in_db, 0.20)
    input_p90_db = _percentile(in_db, 0.90)
    active_threshold_db = max(max(input_floor_db + f32(6.0), input_p90_db - f32(24.0)), f32(-60.0))
    active = in_db >= active_threshold_db
    active_comp = np.maximum(comp[active], f32(0.0))
    active_dees = np.maximum(dees[active], f32(0.0))
    if active_comp.size < 3:
        active_comp = np.maximum(comp, f32(0.0))
        active_dees = np.maximum(dees, f32(0.0))
    active_block_count = int(active_comp.size)
    active_ratio = f32(np.count_nonzero(active_comp >= f32(0.10))) / f32(active_block_count) if active_block_count else f32(0.0)
    audible = in_db > f32(-100.0)
    active_output_gain = _percentile((out_db - in_db)[active & audible], 0.50)
    silence_level_delta = _percentile((out_db - in_db)[(~active) & audible], 0.50)
    silence_output_gain = _percentile(-np.maximum(comp[~active], f32(0.0)), 0.50)
    pumping = _pumping_score(np.maximum(comp, f32(0.0)), 50.0)
    return {
        "input_sample_peak_db": float(_linear_to_db(input_sample_peak)),
        "input_rms_db": float(_linear_to_db(input_rms)),
        "output_sample_peak_db": float(output_sample_peak_db),
        "pre_limiter_true_peak_db": float(pre_limiter_true_peak_db),
        "output_true_peak_db": float(output_true_peak_db),
        "output_rms_db": float(_linear_to_db(output_rms)),
        "limiter_effective_ceiling_db": float(ceiling),
        "sample_headroom_db": float(ceiling - output_sample_peak_db),
        "pre_limiter_true_peak_headroom_db": float(ceiling - pre_limiter_true_peak_db),
        "true_peak_headroom_db": float(ceiling - output_true_peak_db),
        "limiter_gain_reduction_db": float(max(f32(0.0), fmax("limiter_peak_gain_reduction_db"))),
        "true_peak_limiter_gain_reduction_db": float(max(f32(0.0), fmax("true_peak_limiter_gain_reduction_db"))),
        "true_peak_limited_events": int(rows["true_peak_limited_events"].sum()),
        "compressor_gain_reduction_db": float(max(f32(0.0), fmax("compressor_gain_reduction_db"))),
        "deesser_gain_reduction_db": float(max(f32(0.0), fmax("deesser_gain_reduction_db"))),
        "compressor_gain_reduction_median_db": float(_percentile(active_comp, 0.50)),
        "compressor_gain_reduction_p95_db": float(_percentile(active_comp, 0.95)),
        "compressor_gain_reduction_active_ratio": float(active_ratio),
        "active_output_gain_db": float(active_output_gain),
        "silence_output_gain_db": float(silence_output_gain),
        "silence_level_delta_db": float(silence_level_delta),
        "compressor_pumping_score_db": float(pumping),
        "non_finite_output": bool(rows["non_finite_output"].any()),
        "deesser_gain_reduction_median_db": float(_percentile(active_dees, 0.50)),
        "deesser_gain_reduction_p95_db": float(_percentile(active_dees, 0.95)),
        "analysis_block_ms": 20.0,
        "active_analysis_threshold_db": float(active_threshold_db),
        "active_analysis_block_count": active_block_count,
        "processed_samples": samples,
    }


# ------------------------------------------------------------ simulate_auto_eq_chain
def configure_auto_eq_chain(engine: Engine, sample_rate: float, bands, settings: Mapping[str, object] | None) -> float:
    """Replay python_api.rs:400-487 on `engine`; returns the effective limiter ceiling (dB, f32)."""
    engine.set_eq_enabled(1)
    if settings is not None and settings.get("eq_bands_v2") is not None:
        arr = _bands_v2(list(settings["eq_bands_v2"]), sample_rate)
        lib = _lib.load()
        for index in range(NUM_BANDS):
            _lib.check(lib.af_eq_set_band_config(engine._h, index, C.byref(arr[index])))
        engine.eq_reset()
    else:
        for index, (frequency, gain_db, q) in enumerate(bands):
            engine.eq_set_band_frequency(index, float(frequency))
            engine.eq_set_band_gain(index, float(gain_db))
            engine.eq_set_band_q(index, float(q))
    deesser_enabled = _get(settings, "deesser_enabled", False)
    engine.set_eq_before_deesser(int(_get(settings, "eq_before_deesser", False)))
    engine.set_deesser_enabled(int(deesser_enabled))
    if deesser_enabled:
        engine.deesser_set_auto_enabled(int(_get(settings, "deesser_auto_enabled", True)))
        engine.deesser_set_auto_amount(_get(settings, "deesser_auto_amount", 0.5))
        engine.deesser_set_low_cut_hz(_get(settings, "deesser_low_cut_hz", 4000.0))
        engine.deesser_set_high_cut_hz(_get(settings, "deesser_high_cut_hz", 11_000.0))
        engine.deesser_set_threshold_db(_get(settings, "deesser_threshold_db", -28.0))
        engine.deesser_set_ratio(_get(settings, "deesser_ratio", 4.0))
        engine.deesser_set_attack_ms(_get(settings, "deesser_attack_ms", 2.0))
        engine.deesser_set_release_ms(_get(settings, "deesser_release_ms", 80.0))
        engine.deesser_set_max_reduction_db(_get(settings, "deesser_max_reduction_db", 6.0))
    compressor_enabled = _get(settings, "compressor_enabled", True)
    engine.set_compressor_enabled(int(compressor_enabled))
    if compressor_enabled:
        engine.compressor_set_threshold(_get(settings, "compressor_threshold_db", -20.0))
        engine.compressor_set_ratio(_get(settings, "compressor_ratio", 4.0))
        engine.compressor_set_attack_time(_get(settings, "compressor_attack_ms", 10.0))
        engine.compressor_set_release_time(_get(settings, "compressor_release_ms", 200.0))
        engine.compressor_set_makeup_gain(_get(settings, "compressor_makeup_gain_db", 0.0))
        engine.compressor_set_adaptive_release(int(_get(settings, "compressor_adaptive_release", False)))
        engine.compressor_set_base_release_time(_get(settings, "compressor_base_release_ms", 50.0))
        engine.compressor_set_auto_makeup_enabled(int(_get(settings, "compressor_auto_makeup_enabled", False)))
        engine.compressor_set_target_lufs(_get(settings, "compressor_target_lufs", -18.0))
        engine.compressor_set_sidechain_highpass_enabled(int(_get(settings, "compressor_sidechain_highpass_enabled", True)))
    limiter_enabled = _get(settings, "limiter_enabled", True)
    engine.set_limiter_enabled(int(limiter_enabled))
    ceiling_db = _get(settings, "limiter_ceiling_db", -0.5)
    careful = _get(settings, "limiter_careful_output_enabled", True)
    # effective_limiter_ceiling_db, audio/processor/control.rs:904-910, then `as f32`
    effective = float(f32(min(ceiling_db, CAREFUL_OUTPUT_CEILING_DB) if careful else ceiling_db))
    if limiter_enabled:
        release_ms = _get(settings, "limiter_release_ms", 50.0)
        engine.limiter_set_lookahead_ms(_get(settings, "limiter_lookahead_ms", 2.0))
        engine.limiter_set_ceiling(effective)
        engine.limiter_set_release_time(release_ms)
        engine.true_peak_limiter_set_release_ms(float(f32(release_ms)))
    block = int(min(max(round(sample_rate * 0.020), 1), 8192))  # python_api.rs:512-514
    engine.set_control_block_samples(block)
    return effective


def _block_lengths(n: int, block: int) -> np.ndarray:
    full, rest = divmod(n, block)
    return np.asarray([block] * full + ([rest] if rest else []), dtype=np.int64)


GROUP_STREAMS = 64  # streams of one chain workgroup = the granularity of a preset


def _is_preset_list(bands, settings) -> bool:
    return isinstance(settings, (list, tuple)) or (len(bands) > 0 and isinstance(bands[0], (list, tuple)) and len(bands[0]) > 0
                                                    and isinstance(bands[0][0], (list, tuple)))


def simulate_auto_eq_chain_batch(audio: np.ndarray, sample_rate: float, bands, settings=None, device: int = 0,
                                 diagnostics: bool = True) -> tuple[np.ndarray, list[dict[str, Any]]]:
    """Batched form: ``audio`` is [n_streams, n] float32.

    ``bands`` / ``settings`` are either one preset for every stream (the reference's arguments), or one entry per
    stream (a list of 10-band lists / a list of settings dicts): the reference configures one processor per stream
    (python/mic_eq/config_parts/settings.py:543-593).  Streams that share a preset are packed into groups of 64 (the unit
    a preset applies to on the GPU; partial groups are padded with silent streams) and handed back in the caller's order.

    Returns (output [n_streams, n] float32, one reference-shaped dict per stream).
    """
    started = time.perf_counter()
    if not np.isfinite(sample_rate) or sample_rate <= 0.0:
        raise ValueError("sample_rate must be positive and finite")
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    if audio.ndim != 2:
        raise TypeError("audio must be [n_streams, n_samples]")
    n_streams, n = audio.shape
    if not _is_preset_list(bands, settings):
        if len(bands) != NUM_BANDS:
            raise ValueError(f"expected {NUM_BANDS} EQ bands, got {len(bands)}")
        presets = [(bands, settings)]
        stream_preset = np.zeros(n_streams, dtype=np.int64)
    else:
        per_bands = list(bands) if (len(bands) and isinstance(bands[0][0], (list, tuple))) else [bands] * n_streams
        per_settings = list(settings) if isinstance(settings, (list, tuple)) else [settings] * n_streams
        if len(per_bands) != n_streams or len(per_settings) != n_streams:
            raise ValueError(f"expected one preset per stream ({n_streams}), got {len(per_bands)} band lists and {len(per_settings)} settings")
        presets, index, stream_preset = [], {}, np.zeros(n_streams, dtype=np.int64)
        for s_i, (b, st) in enumerate(zip(per_bands, per_settings)):
            if len(b) != NUM_BANDS:
                raise ValueError(f"expected {NUM_BANDS} EQ bands, got {len(b)}")
            key = (repr([tuple(x) for x in b]), repr(sorted((st or {}).items(), key=lambda kv: kv[0])))
            if key not in index:
                index[key] = len(presets)
                presets.append((b, st))
            stream_preset[s_i] = index[key]
    # pack: streams of preset k -> whole groups of 64
    order, group_preset = [], []
    for k in range(len(presets)):
        members = np.flatnonzero(stream_preset == k)
        padded = -(-members.size // GROUP_STREAMS) * GROUP_STREAMS if len(presets) > 1 else members.size
        order.extend(members.tolist() + [-1] * (padded - members.size))
        group_preset.extend([k] * (-(-padded // GROUP_STREAMS)))
    order = np.asarray(order, dtype=np.int64)
    packed = np.zeros((order.size, n), dtype=np.float32)
    packed[order >= 0] = audio[order[order >= 0]]
    engine = Engine(sample_rate, int(order.size), device)
    try:
        effective = []
        if len(presets) > 1:
            engine.set_preset_count(len(presets))
        for k, (b, st) in enumerate(presets):
            if len(presets) > 1:
                engine.select_preset(k)
            effective.append(configure_auto_eq_chain(engine, float(sample_rate), b, st))
        if len(presets) > 1:
            gp = np.asarray(group_preset, dtype=np.int32)
            _lib.check(engine._lib.af_engine_assign_presets(engine._h, gp.ctypes.data_as(C.POINTER(C.c_int32)), int(gp.size)))
        packed_out = engine.process(packed) if n else packed.copy()
        rows = engine.block_stats()
    finally:
        engine.close()
    output = np.empty_like(audio)
    position = np.empty(n_streams, dtype=np.int64)
    position[order[order >= 0]] = np.flatnonzero(order >= 0)
    output[:] = packed_out[position]
    results: list[dict[str, Any]] = []
    if diagnostics:
        block = int(min(max(round(sample_rate * 0.020), 1), 8192))
        lengths = _block_lengths(n, block)
        runtime_ms = (time.perf_counter() - started) * 1000.0
        for s_i in range(n_streams):
            d = chain_diagnostics(rows[:, position[s_i]], lengths, effective[int(stream_preset[s_i])])
            d["candidate_runtime_ms"] = runtime_ms
            results.append(d)
    return output, results


def simulate_auto_eq_chain(audio, sample_rate: float, bands: Sequence[tuple[float, float, float]],
                           settings: Mapping[str, object] | None = None) -> dict[str, Any]:
    """python_api.rs:378-714 -- de-esser -> EQ -> compressor -> limiter -> true-peak limiter."""
    started = time.perf_counter()
    if not np.isfinite(sample_rate) or sample_rate <= 0.0:
        raise ValueError("sample_rate must be positive and finite")
    if len(bands) != NUM_BANDS:
        raise ValueError(f"expected {NUM_BANDS} EQ bands, got {len(bands)}")
    audio = _audio_1d(audio)
    output, results = simulate_auto_eq_chain_batch(audio.reshape(1, -1), sample_rate, bands, settings)
    d = results[0]
    d["candidate_runtime_ms"] = (time.perf_counter() - started) * 1000.0
    if _get(settings, "return_output_audio", False):
        d["output_audio"] = output[0].tolist()  # the reference returns a Python list of floats
    return d


# ----------------------------------------------------------------------- suppressor
TIMING_REPETITIONS = 7  # python/tools/evaluate_limiter_lookahead.py:28 (1 warm-up + 7 timed runs)


def _percentile_f64(values, q: float) -> float:
    v = np.sort(np.asarray(values, dtype=np.float64))
    if v.size == 0:
        return 0.0
    pos = (v.size - 1) * min(max(q, 0.0), 1.0)
    lo, hi = int(np.floor(pos)), int(np.ceil(pos))
    return float(v[lo] + (pos - lo) * (v[hi] - v[lo]))


def suppress(audio: np.ndarray, strength: float = 1.0, weight_seed: int | None = None, raw_protocol: bool = False,
             device: int = 0, kernel_ms: list | None = None) -> np.ndarray:
    """RNNoiseProcessor::process_frames (rust-core/src/dsp/rnnoise.rs:122-164) over [n_streams, n] or [n] audio.

    Only whole 480-sample frames are produced (a shorter tail stays "buffered", as in the reference).
    `raw_protocol=True` is the scaling of bin/rnnoise_benchmark.rs (clamp(+-1)*32768, no wet/dry mix).
    `kernel_ms` (optional list) receives the HIP-event time of the call's kernels.
    """
    a = np.ascontiguousarray(audio, dtype=np.float32)
    squeeze = a.ndim == 1
    if squeeze:
        a = a.reshape(1, -1)
    frames = a.shape[1] // 480
    a = np.ascontiguousarray(a[:, : frames * 480])
    if frames == 0:
        return a[0] if squeeze else a
    engine = Engine(48_000.0, a.shape[0], device)
    try:
        engine.set_eq_enabled(0)
        engine.set_compressor_enabled(0)
        engine.set_limiter_enabled(0)
        engine.set_suppressor_enabled(1)
        engine.set_suppressor_strength(float(strength))
        engine.suppressor_set_raw_protocol(int(raw_protocol))
        if weight_seed is not None:
            engine.suppressor_set_synthetic_weights(int(weight_seed))
        engine.set_control_block_samples(480)
        if kernel_ms is not None:
            engine.set_timing_enabled(1)
        out = engine.process(a)
        if kernel_ms is not None:
            kernel_ms.append(engine.last_kernel_ms()[0])
    finally:
        engine.close()
    return out[0] if squeeze else out


# ------------------------------------------------- NoiseSuppressor (noise_suppressor.rs:18-194)
class NoiseModel:
    """NoiseModel (noise_suppressor.rs:18-87): ids, display names, the models a build offers."""

    RNNOISE, DEEPFILTER_LL, DEEPFILTER = 0, 1, 2

    @staticmethod
    def from_id(model_id: str):
        model = C.c_int32(0)
        rc = _lib.load().af_noise_model_from_id(str(model_id).encode(), C.byref(model))
        return model.value if rc == 0 else None  # Option<NoiseModel>

    @staticmethod
    def id(model: int) -> str:
        return _lib.load().af_noise_model_id(int(model)).decode()

    @staticmethod
    def display_name(model: int) -> str:
        return _lib.load().af_noise_model_display_name(int(model)).decode()

    @staticmethod
    def available() -> list[int]:
        buf = (C.c_int32 * 4)()
        n = _lib.load().af_noise_model_available(buf, 4)
        return [int(buf[i]) for i in range(n)]


class NoiseSuppressor:
    """The `NoiseSuppressor` trait (noise_suppressor.rs:89-165) over a batch of streams that advance in lock step:
    push_samples -> process_frames -> pop_samples_into, with RNNoiseProcessor's two fixed rings (rnnoise.rs:11)."""

    def __init__(self, model: int | str = "rnnoise", n_streams: int = 1, device: int = 0, weight_seed: int | None = None):
        self._lib = _lib.load()
        if isinstance(model, str):
            parsed = NoiseModel.from_id(model)
            if parsed is None:
                raise ValueError(f"unknown noise model id {model!r}")
            model = parsed
        handle = C.c_void_p()
        _lib.check(self._lib.af_noise_suppressor_create(int(model), int(n_streams), int(device), C.byref(handle)))
        self._h = handle
        self.n_streams = int(n_streams)
        if weight_seed is not None:
            _lib.check(self._lib.af_suppressor_set_synthetic_weights(self._lib.af_noise_suppressor_engine(self._h), int(weight_seed)))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.af_noise_suppressor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _count(self, value: int) -> int:
        if value < 0:
            _lib.check(int(value))
        return int(value)

    def push_samples(self, samples: np.ndarray) -> int:
        a = np.ascontiguousarray(samples, dtype=np.float32)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        if a.shape[0] != self.n_streams:
            raise ValueError(f"expected {self.n_streams} streams, got {a.shape[0]}")
        return self._count(self._lib.af_noise_suppressor_push_samples(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[1], a.shape[1]))

    def process_frames(self) -> None:
        _lib.check(self._lib.af_noise_suppressor_process_frames(self._h))

    def available_samples(self) -> int:
        return int(self._lib.af_noise_suppressor_available_samples(self._h))

    def pending_input(self) -> int:
        return int(self._lib.af_noise_suppressor_pending_input(self._h))

    def pop_samples(self, count: int) -> np.ndarray:
        count = max(int(count), 0)
        out = np.zeros((self.n_streams, max(count, 1)), dtype=np.float32)
        n = self._count(self._lib.af_noise_suppressor_pop_samples_into(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), count, out.shape[1]))
        return np.ascontiguousarray(out[:, :n])

    def pop_all_samples(self) -> np.ndarray:
        return self.pop_samples(self.available_samples())

    def drain_pending_input(self) -> np.ndarray:
        cap = 8192 + 480
        out = np.zeros((self.n_streams, cap), dtype=np.float32)
        n = self._count(self._lib.af_noise_suppressor_drain_pending_input(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), cap, cap))
        return np.ascontiguousarray(out[:, :n])

    def set_strength(self, value: float) -> None:
        _lib.check(self._lib.af_noise_suppressor_set_strength(self._h, float(value)))

    def get_strength(self) -> float:
        return float(self._lib.af_noise_suppressor_get_strength(self._h))

    def set_enabled(self, enabled: bool) -> None:
        _lib.check(self._lib.af_noise_suppressor_set_enabled(self._h, int(bool(enabled))))

    def is_enabled(self) -> bool:
        return bool(self._lib.af_noise_suppressor_is_enabled(self._h))

    def soft_reset(self) -> None:
        _lib.check(self._lib.af_noise_suppressor_soft_reset(self._h))

    def reset(self) -> None:
        _lib.check(self._lib.af_noise_suppressor_reset(self._h))

    def model_type(self) -> int:
        return int(self._lib.af_noise_suppressor_model_type(self._h))

    def latency_samples(self) -> int:
        return int(self._lib.af_noise_suppressor_latency_samples(self._h))

    def backend_available(self) -> bool:
        return bool(self._lib.af_noise_suppressor_backend_available(self._h))

    def backend_failed(self) -> bool:
        return bool(self._lib.af_noise_suppressor_backend_failed(self._h))

    def backend_error(self):
        msg = self._lib.af_noise_suppressor_backend_error(self._h)
        return msg.decode() if msg else None


def new_noise_suppression_engine(model: int | str, n_streams: int = 1, device: int = 0) -> NoiseSuppressor:
    """new_noise_suppression_engine (noise_suppressor.rs:168-194)."""
    return NoiseSuppressor(model, n_streams, device)


def rnnoise_benchmark(input_path: str, output_path: str, metadata_path: str, weight_seed: int | None = None) -> dict:
    """File protocol of rust-core/src/bin/rnnoise_benchmark.rs:51-117 (`<in.f32> <out.f32> <meta.json>`),
    the boundary python/tools/evaluate_rnnoise_backends.py:106-125 drives."""
    import json

    data = np.fromfile(input_path, dtype="<f4")
    n = data.size
    frames = -(-n // 480)
    padded = np.zeros(frames * 480, dtype=np.float32)
    padded[:n] = data
    # The reference times every frame on the CPU (rnnoise_benchmark.rs:80-96).  A batched engine runs all frames of the
    # file in one pipelined call, so a per-frame clock does not exist; what is measured instead is the whole call, 1 warm-up
    # + TIMING_REPETITIONS fresh runs (HIP events around the call's kernels), each divided by the frame count: the
    # percentiles below are percentiles of those per-run frame times, and `frame_time_basis` says so.
    kernel_ms: list[float] = []
    out = suppress(padded, 1.0, weight_seed, raw_protocol=True) if frames else padded  # warm-up; its output is the result
    for _ in range(TIMING_REPETITIONS if frames else 0):
        suppress(padded, 1.0, weight_seed, raw_protocol=True, kernel_ms=kernel_ms)
    out[:n].astype("<f4").tofile(output_path)
    per_frame = [ms / 1000.0 / frames for ms in kernel_ms] if frames else [0.0]
    elapsed = _percentile_f64(kernel_ms, 0.5) / 1000.0 if frames else 0.0
    meta = {
        "frames": int(frames), "samples": int(n), "elapsed_seconds": elapsed,
        "rtf": elapsed / max(n / 48_000.0, 1e-12),
        "frame_p95_seconds": _percentile_f64(per_frame, 0.95), "frame_p99_seconds": _percentile_f64(per_frame, 0.99),
        "frame_max_seconds": float(max(per_frame)),
        "frame_time_basis": f"whole-call GPU time / frames, percentiles over {TIMING_REPETITIONS} repetitions (median = elapsed_seconds)",
        "repetitions": TIMING_REPETITIONS if frames else 0,
    }
    with open(metadata_path, "w", encoding="utf-8") as fh:
        json.dump(meta, fh)
    return meta


# ------------------------------------------------------ simulate_auto_makeup_control
def simulate_auto_makeup_control(audio, sample_rate: float, vad_probabilities: Sequence[float], noise_floor_db: float,
                                 noise_reliability: float, settings: Mapping[str, object] | None = None) -> dict[str, Any]:
    """python_api.rs:118-276: the compressor's auto-makeup controller at the 10 ms control cadence."""
    control_block = 480
    if not np.isfinite(sample_rate) or sample_rate <= 0.0:
        raise ValueError("sample_rate must be positive and finite")
    if not np.isfinite(noise_floor_db) or not np.isfinite(noise_reliability) or not 0.0 <= noise_reliability <= 1.0:
        raise ValueError("noise evidence must be finite and reliability must be between 0 and 1")
    vad = np.ascontiguousarray(vad_probabilities, dtype=np.float64)
    if vad.size and (not np.all(np.isfinite(vad)) or vad.min() < 0.0 or vad.max() > 1.0):
        raise ValueError("VAD probabilities must be finite and between 0 and 1")
    audio = _audio_1d(audio)
    n = audio.size
    block_count = -(-n // control_block)
    if vad.size and vad.size != block_count:
        raise ValueError(f"expected {block_count} VAD probabilities at the 10 ms control cadence, got {vad.size}")
    vad_reliability = _get(settings, "vad_reliability", 1.0)
    if not np.isfinite(vad_reliability) or not 0.0 <= vad_reliability <= 1.0:
        raise ValueError("vad_reliability must be finite and between 0 and 1")
    run_ms: list[float] = []

    def run_once():
        engine = Engine(sample_rate, 1)
        try:
            return _auto_makeup_run(engine, audio, n, sample_rate, vad, vad_reliability, noise_floor_db, noise_reliability, settings,
                                    control_block, run_ms)
        finally:
            engine.close()

    output, rows = run_once()  # warm-up; its output is the result
    for _ in range(TIMING_REPETITIONS if n else 0):
        run_once()
    run_ms = run_ms[1:] if len(run_ms) > 1 else run_ms
    lengths = _block_lengths(n, control_block).astype(np.float64)
    # The reference clocks every 480-sample block on the CPU (python_api.rs:203-240).  Here all blocks of the clip run in
    # one launch pair, so the figures are per-run block times (run time / blocks) over TIMING_REPETITIONS fresh runs.
    per_block_ms = [ms / max(block_count, 1) for ms in run_ms] or [0.0]
    result: dict[str, Any] = {
        "control_block_size": control_block,
        "control_cadence_hz": sample_rate / control_block,
        "processed_samples": int(n),
        "makeup_gain_db": rows["compressor_makeup_gain_db"].astype(np.float32).tolist(),
        "activity": rows["auto_makeup_activity"].astype(np.float32).tolist(),
        "reliability": rows["auto_makeup_reliability"].astype(np.float32).tolist(),
        "gain_reduction_db": rows["compressor_gain_reduction_db"].astype(np.float32).tolist(),
        "input_rms_db": _linear_to_db(np.sqrt(rows["input_square_sum"] / np.maximum(lengths, 1.0)).astype(np.float32)).tolist(),
        "output_rms_db": _linear_to_db(np.sqrt(rows["output_square_sum"] / np.maximum(lengths, 1.0)).astype(np.float32)).tolist(),
        "p95_block_runtime_ms": _percentile_f64(per_block_ms, 0.95),
        "p99_block_runtime_ms": _percentile_f64(per_block_ms, 0.99),
        "max_block_runtime_ms": float(max(per_block_ms)),
        "block_runtime_basis": f"whole-clip run time / blocks, percentiles over {TIMING_REPETITIONS} repetitions",
    }
    if _get(settings, "return_output_audio", False):
        result["output_audio"] = output.tolist()
    return result


def _auto_makeup_run(engine, audio, n, sample_rate, vad, vad_reliability, noise_floor_db, noise_reliability, settings,
                     control_block, run_ms):
    """One fresh run of the controller (python_api.rs:150-242 setter sequence); appends its wall-clock ms to `run_ms`."""
    engine.set_eq_enabled(0)
    engine.set_limiter_enabled(0)
    engine.set_compressor_enabled(1)
    engine.compressor_set_threshold(_get(settings, "threshold_db", -24.0))
    engine.compressor_set_ratio(_get(settings, "ratio", 3.0))
    engine.compressor_set_attack_time(_get(settings, "attack_ms", 10.0))
    engine.compressor_set_release_time(_get(settings, "release_ms", 180.0))
    engine.compressor_set_makeup_gain(_get(settings, "makeup_gain_db", 0.0))
    engine.compressor_set_auto_makeup_enabled(1)
    engine.compressor_set_target_lufs(_get(settings, "target_lufs", -18.0))
    engine.compressor_set_noise_reference_reliability(float(noise_reliability))
    engine.compressor_set_adaptive_release(int(_get(settings, "adaptive_release", True)))
    engine.compressor_set_sidechain_highpass_enabled(int(_get(settings, "sidechain_highpass_enabled", True)))
    engine.set_control_block_samples(control_block)
    engine.set_input_scrub_enabled(0)
    dp = C.POINTER(C.c_double)
    _lib.check(engine._lib.af_compressor_set_activity_evidence(
        engine._h, vad.ctypes.data_as(dp), int(vad.size), 0, float(vad_reliability), float(noise_floor_db),
        float(noise_reliability)))
    started = time.perf_counter()
    output = engine.process(audio.reshape(1, -1))[0] if n else audio.copy()
    run_ms.append((time.perf_counter() - started) * 1000.0)
    rows = engine.block_stats()[:, 0]
    return output, rows


# -------------------------------------------------------------------- simulate_eq_v2
def simulate_eq_v2(audio, sample_rate: float, bands, return_output_audio: bool = False) -> dict[str, Any]:
    """lib.rs:214-288: typed 10-band EQ over the whole clip, peaks / true peaks / rms."""
    arr = _bands_v2(list(bands), float(sample_rate))
    audio = _audio_1d(audio)
    if not np.all(np.isfinite(audio)):
        raise ValueError("audio must contain only finite samples")
    lib = _lib.load()
    n = audio.size

    def run(eq_enabled: bool) -> tuple[np.ndarray, np.ndarray, Engine]:
        engine = Engine(sample_rate, 1)
        engine.set_limiter_enabled(0)
        engine.set_compressor_enabled(0)
        engine.set_eq_enabled(int(eq_enabled))
        if eq_enabled:
            for index in range(NUM_BANDS):
                _lib.check(lib.af_eq_set_band_config(engine._h, index, C.byref(arr[index])))
            engine.eq_reset()
        engine.set_control_block_samples(8192)
        out = engine.process(audio.reshape(1, -1))[0] if n else audio.copy()
        return out, engine.block_stats()[:, 0], engine

    started = time.perf_counter()
    output, rows, engine = run(True)
    runtime_ms = (time.perf_counter() - started) * 1000.0
    freqs = 20.0 * np.power(20_000.0 / 20.0, np.arange(512, dtype=np.float64) / 511.0)
    response = np.zeros(512, dtype=np.float64)
    dp = C.POINTER(C.c_double)
    _lib.check(lib.af_engine_eq_magnitude_response(engine._h, freqs.ctypes.data_as(dp), 512, response.ctypes.data_as(dp)))
    engine.close()
    # the input true peak is the same detector over the untouched clip (lib.rs:257-260)
    _, in_rows, probe = run(False)
    probe.close()
    divisor = float(max(n, 1))
    in_sq = float(np.sum(in_rows["input_square_sum"])) if n else 0.0
    out_sq = float(np.sum(rows["output_square_sum"])) if n else 0.0
    result: dict[str, Any] = {
        "input_sample_peak": float(in_rows["input_sample_peak"].max()) if n else 0.0,
        "output_sample_peak": float(rows["output_sample_peak"].max()) if n else 0.0,
        "input_true_peak": float(in_rows["output_true_peak"].max()) if n else 0.0,
        "output_true_peak": float(rows["output_true_peak"].max()) if n else 0.0,
        "input_rms": float(np.sqrt(in_sq / divisor)),
        "output_rms": float(np.sqrt(out_sq / divisor)),
        "max_response_db": float(response.max()),
        "runtime_ms": runtime_ms,
        "sample_count": int(n),
        "algorithmic_latency_samples": 0,
        "non_finite_output": bool(not np.all(np.isfinite(output))),
    }
    if return_output_audio:
        result["output_audio"] = output.tolist()
    return result


# ------------------------------------------------------------- eq_magnitude_response
def eq_magnitude_response(frequencies_hz: Sequence[float], bands: Sequence[tuple[float, float, float]],
                          sample_rate: float) -> list[float]:
    """lib.rs:99-150."""
    if len(bands) != NUM_BANDS:
        raise ValueError(f"expected {NUM_BANDS} EQ bands, got {len(bands)}")
    f = np.ascontiguousarray(frequencies_hz, dtype=np.float64)
    b = np.ascontiguousarray(np.asarray(bands, dtype=np.float64).reshape(NUM_BANDS, 3))
    out = np.zeros_like(f)
    dp = C.POINTER(C.c_double)
    _lib.check(_lib.load().af_eq_magnitude_response(f.ctypes.data_as(dp), f.size, b.ctypes.data_as(dp), float(sample_rate),
                                                     out.ctypes.data_as(dp)))
    return out.tolist()


def eq_magnitude_response_v2(frequencies_hz: Sequence[float], bands, sample_rate: float) -> list[float]:
    """lib.rs:191-212."""
    arr = _bands_v2(list(bands), float(sample_rate))
    f = np.ascontiguousarray(frequencies_hz, dtype=np.float64)
    out = np.zeros_like(f)
    dp = C.POINTER(C.c_double)
    _lib.check(_lib.load().af_eq_magnitude_response_v2(f.ctypes.data_as(dp), f.size, arr, float(sample_rate),
                                                        out.ctypes.data_as(dp)))
    return out.tolist()


# ------------------------------------------------------------------ product resampler
RESAMPLER_CHUNK_SIZE = 1024            # audio/processor.rs:53
PRODUCT_RESAMPLER_SINC_LEN = 128       # audio/processor.rs:54
PRODUCT_RESAMPLER_WINDOW_NAME = "blackman"  # audio/processor.rs:55
RESAMPLER_WINDOW_IDS = {"blackman_harris": 0, "blackman_harris_squared": 1, "blackman": 2, "blackman_squared": 3,
                        "hann": 4, "hann_squared": 5}  # resampler_window_from_name, resampling.rs:158-168


def product_resampler_configuration() -> tuple[int, str, str, int, int]:
    """resampling.rs:263-272."""
    return (PRODUCT_RESAMPLER_SINC_LEN, PRODUCT_RESAMPLER_WINDOW_NAME, "cubic", 256, RESAMPLER_CHUNK_SIZE)


class Resampler:
    """One resampling plan (rates, chunk size, sinc length, window) for any number of equally long streams."""

    def __init__(self, input_rate: int, output_rate: int, chunk_size: int = RESAMPLER_CHUNK_SIZE,
                 sinc_len: int | None = None, window: str | None = None, device: int = 0):
        # argument checks in the order of resampling.rs:187-214
        if int(input_rate) <= 0 or int(output_rate) <= 0:
            raise ValueError("sample rates must be positive")
        if not 1 <= int(chunk_size) <= RESAMPLER_CHUNK_SIZE:
            raise ValueError(f"chunk_size must be between 1 and {RESAMPLER_CHUNK_SIZE}")
        sinc_len = PRODUCT_RESAMPLER_SINC_LEN if sinc_len is None else int(sinc_len)
        if not 32 <= sinc_len <= 2048 or sinc_len & (sinc_len - 1):
            raise ValueError("sinc_len must be a power of two between 32 and 2048")
        name = PRODUCT_RESAMPLER_WINDOW_NAME if window is None else window
        if name not in RESAMPLER_WINDOW_IDS:
            raise ValueError(f"unsupported resampler window {name!r}")
        self._lib = _lib.load()
        handle = C.c_void_p()
        _lib.check(self._lib.af_resampler_create(int(input_rate), int(output_rate), int(chunk_size), sinc_len,
                                                 RESAMPLER_WINDOW_IDS[name], int(device), C.byref(handle)))
        self._h = handle
        self.input_rate, self.output_rate, self.sinc_len = int(input_rate), int(output_rate), sinc_len

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.af_resampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def output_delay(self) -> int:
        return int(self._lib.af_resampler_output_delay(self._h))

    def expected_frames(self, n_in: int) -> int:
        return int(self._lib.af_resampler_expected_frames(self._h, int(n_in)))

    def plan(self, n_in: int) -> tuple[int, int]:
        """(frames the reference's driver loop returns for n_in input frames, number of chunks it runs)."""
        n_out, blocks = C.c_int64(0), C.c_int64(0)
        _lib.check(self._lib.af_resampler_plan(self._h, int(n_in), C.byref(n_out), C.byref(blocks)))
        return n_out.value, blocks.value

    def sinc_table(self) -> np.ndarray:
        table = np.zeros((256, int(self._lib.af_resampler_sinc_len(self._h))), dtype=np.float64)
        _lib.check(self._lib.af_resampler_copy_sinc_table(self._h, table.ctypes.data_as(C.POINTER(C.c_double))))
        return table

    def process(self, samples: np.ndarray) -> np.ndarray:
        """[n_streams, n_in] (or [n_in]) float64 host array -> [n_streams, n_out]."""
        x = np.ascontiguousarray(samples, dtype=np.float64)
        squeeze = x.ndim == 1
        if squeeze:
            x = x.reshape(1, -1)
        n_streams, n_in = x.shape
        n_out, _ = self.plan(n_in)
        out = np.zeros((n_streams, n_out), dtype=np.float64)
        dp = C.POINTER(C.c_double)
        _lib.check(self._lib.af_resampler_process_host(self._h, x.ctypes.data_as(dp), out.ctypes.data_as(dp), n_in, n_streams,
                                                       max(n_in, 1), max(n_out, 1) if n_out == 0 else n_out))
        return out[0] if squeeze else out

    def process_device(self, in_ptr: int, out_ptr: int, n_in: int, n_streams: int, in_stride: int, out_stride: int,
                       hip_stream: int = 0) -> None:
        _lib.check(self._lib.af_resampler_process_device(self._h, C.c_void_p(in_ptr), C.c_void_p(out_ptr), int(n_in),
                                                         int(n_streams), int(in_stride), int(out_stride), C.c_void_p(hip_stream)))

    def last_kernel_ms(self) -> float:
        ms = C.c_double(0.0)
        _lib.check(self._lib.af_resampler_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value


def simulate_product_resampler_batch(samples: np.ndarray, input_rate: int, output_rate: int,
                                     chunk_size: int = RESAMPLER_CHUNK_SIZE, sinc_len: int | None = None,
                                     window: str | None = None, repetitions: int = 0):
    """Batched form: samples [n_streams, n] -> (output [n_streams, n_out] f64, delay, expected_frames, blocks, kernel ms).

    `repetitions` > 0 re-runs the launch that many times and returns the list of their HIP-event times instead of one."""
    r = Resampler(input_rate, output_rate, chunk_size, sinc_len, window)
    try:
        x = np.asarray(samples, dtype=np.float64)
        if not np.all(np.isfinite(x)):
            raise ValueError("samples must be finite")
        out = r.process(x)
        _, blocks = r.plan(x.shape[-1])
        kernel_ms = [r.last_kernel_ms()]
        for _ in range(int(repetitions)):  # the resampler is stateless across calls: re-running is a pure timing repeat
            r.process(x)
            kernel_ms.append(r.last_kernel_ms())
        timing = kernel_ms[1:] if repetitions else kernel_ms[0]
        return out, r.output_delay, r.expected_frames(x.shape[-1]), blocks, timing
    finally:
        r.close()


def simulate_product_resampler(samples: Sequence[float], input_rate: int, output_rate: int,
                               chunk_size: int = RESAMPLER_CHUNK_SIZE, sinc_len: int | None = None,
                               window: str | None = None) -> tuple[list[float], int, int, list[int]]:
    """resampling.rs:170-261: (output, delay, expected_frames, block_times_ns).

    The reference clocks each 1024-frame chunk on the CPU; here all chunks run in ONE launch, so a per-chunk clock does
    not exist.  The launch is timed TIMING_REPETITIONS times (HIP events, after one warm-up); entry i of block_times_ns is
    repetition (i mod TIMING_REPETITIONS)'s time divided by the number of chunks, so the percentiles the harness takes over
    the list (evaluate_resampler_quality.py `runtime` block) are percentiles over real, separately timed launches."""
    x = np.ascontiguousarray(samples, dtype=np.float64).reshape(-1)
    out, delay, expected, blocks, kernel_ms = simulate_product_resampler_batch(x.reshape(1, -1), input_rate, output_rate,
                                                                             chunk_size, sinc_len, window,
                                                                             repetitions=TIMING_REPETITIONS)
    per_block = [int(round(ms * 1e6 / max(blocks, 1))) for ms in kernel_ms]
    return out[0].tolist(), int(delay), int(expected), [per_block[i % len(per_block)] for i in range(int(blocks))]


# ------------------------------------------------------------------ integrated loudness
LOUDNESS_SAMPLE_RATES = (8000, 16000, 32000, 44100, 48000, 88200, 96000)  # loudness.rs:36-41


def measure_integrated_loudness_batch(audio: np.ndarray, sample_rate: int, device: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """[n_streams, n] float32 -> (LUFS per stream, status per stream: 0 ok, -3 non-finite, -5 nothing passed the gates)."""
    a = np.ascontiguousarray(audio, dtype=np.float32)
    if a.ndim != 2:
        raise ValueError("audio must be [n_streams, n]")
    if int(sample_rate) not in LOUDNESS_SAMPLE_RATES:
        raise ValueError(f"Invalid sample rate: {sample_rate}")
    if a.shape[1] == 0:
        raise ValueError("Invalid audio: at least one sample is required")
    lufs = np.zeros(a.shape[0], dtype=np.float64)
    status = np.zeros(a.shape[0], dtype=np.int32)
    rc = _lib.load().af_measure_integrated_loudness_host(a.ctypes.data_as(C.POINTER(C.c_float)), a.shape[1], a.shape[0], a.shape[1],
                                                         int(sample_rate), int(device), lufs.ctypes.data_as(C.POINTER(C.c_double)),
                                                         status.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc == _lib.AF_ERR_BACKEND:
        _lib.check(rc)
    return lufs, status


def measure_integrated_loudness(audio, sample_rate: int) -> float:
    """lib.rs:290-298: gated BS.1770 loudness (LUFS) of one clip; ValueError like the reference's PyValueError."""
    a = _audio_1d(audio)
    if int(sample_rate) not in LOUDNESS_SAMPLE_RATES:
        raise ValueError(f"Invalid sample rate: {sample_rate}")
    if a.size == 0:
        raise ValueError("Invalid audio: at least one sample is required")
    lufs, status = measure_integrated_loudness_batch(a.reshape(1, -1), sample_rate)
    if status[0] == _lib.AF_ERR_NON_FINITE:
        raise ValueError("Invalid audio: samples must be finite")
    if status[0] != 0:
        raise ValueError("Loudness measurement failed: audio did not produce a finite gated loudness")
    return float(lufs[0])


# ------------------------------------------------------------------ noise gate + suppressor order
RNNOISE_FRAME = 480  # dsp/rnnoise.rs:3


def gate_batch(audio: np.ndarray, threshold_db: float = -40.0, attack_ms: float = 10.0, release_ms: float = 100.0,
               sample_rate: float = 48_000.0, vad_mode: bool = True, trace_block: int = RNNOISE_FRAME, device: int = 0):
    """NoiseGate expander path (gate.rs:626-637) over [n_streams, n]: (output, gain trace [blocks, n_streams], chatter events)."""
    a = np.ascontiguousarray(audio, dtype=np.float32)
    if a.ndim != 2:
        raise ValueError("audio must be [n_streams, n]")
    n_streams, n = a.shape
    out = np.empty_like(a)
    blocks = -(-n // trace_block)
    trace = np.zeros((blocks, n_streams), dtype=np.float32)
    chatter = np.zeros(n_streams, dtype=np.uint64)
    fp = C.POINTER(C.c_float)
    _lib.check(_lib.load().af_gate_process_host(a.ctypes.data_as(fp), out.ctypes.data_as(fp), n, n_streams, n, float(threshold_db),
                                                float(attack_ms), float(release_ms), float(sample_rate), int(bool(vad_mode)),
                                                int(trace_block), trace.ctypes.data_as(fp),
                                                chatter.ctypes.data_as(C.POINTER(C.c_uint64)), int(device)))
    return out, trace, chatter


def simulate_gate_suppressor_order(audio, vad_probabilities: Sequence[float], suppressor_before_gate: bool,
                                   suppressor_strength: float = 1.0, settings: Mapping[str, object] | None = None) -> dict[str, Any]:
    """python_api.rs:288-376: gate (VadAssisted, no VadAutoGate attached) and RNNoise in either order, 480-sample frames."""
    import time

    strength = float(suppressor_strength)
    if not np.isfinite(strength) or not 0.0 <= strength <= 1.0:
        raise ValueError("suppressor_strength must be finite and between 0 and 1")
    x = _audio_1d(audio)
    n = x.size
    frames = -(-n // RNNOISE_FRAME)
    probs = np.asarray(list(vad_probabilities), dtype=np.float64)
    if probs.size != frames or not np.all(np.isfinite(probs)) or np.any(probs < 0.0) or np.any(probs > 1.0):
        raise ValueError(f"expected {frames} finite VAD probabilities at the 10 ms RNNoise cadence")
    started = time.perf_counter()
    padded = np.zeros((1, frames * RNNOISE_FRAME), dtype=np.float32)
    padded[0, :n] = x
    gate_args = (_get(settings, "gate_threshold_db", -40.0), _get(settings, "gate_attack_ms", 10.0),
                 _get(settings, "gate_release_ms", 100.0), 48_000.0, True, RNNOISE_FRAME)
    if frames == 0:
        out, trace, chatter = padded, np.zeros((0, 1), np.float32), np.zeros(1, np.uint64)
    elif suppressor_before_gate:
        out, trace, chatter = gate_batch(suppress(padded, strength), *gate_args)
    else:
        gated, trace, chatter = gate_batch(padded, *gate_args)
        out = suppress(gated, strength)
    return {
        "output_audio": out[0, :n].tolist(),
        "gate_gain": [float(v) for v in trace[:, 0]],
        "gate_chatter_event_count": int(chatter[0]),
        "gate_noise_floor_db": -60.0,            # NoiseGate::noise_floor() without a VadAutoGate, gate.rs:929-934
        "gate_noise_floor_reliability": 0.0,     # gate.rs:938-943
        "suppressor_latency_samples": RNNOISE_FRAME,  # rnnoise.rs:313-315
        "runtime_ms": (time.perf_counter() - started) * 1000.0,
    }
