"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see af_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import pathlib
import subprocess

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = HERE / "libaf_oracle.so"


def build(force: bool = False) -> pathlib.Path:
    src_newer = (not LIB_PATH.exists()) or any(
        (HERE / f).stat().st_mtime > LIB_PATH.stat().st_mtime
        for f in ("af_oracle.c", "af_oracle.h", "tp_fir_table.h", "af_rnnoise.c")
        if (HERE / f).exists()
    )
    if force or src_newer:
        subprocess.run(["make", "-C", str(HERE), "-s"], check=True)
    return LIB_PATH


class EqBandConfig(C.Structure):
    _fields_ = [
        ("filter_type", C.c_int32),
        ("frequency_hz", C.c_double),
        ("gain_db", C.c_double),
        ("q", C.c_double),
        ("slope_db_per_octave", C.c_int32),
        ("enabled", C.c_int32),
    ]


EQ_TYPE_IDS = {"low_shelf": 0, "bell": 1, "high_shelf": 2, "notch": 3, "high_pass": 4, "low_pass": 5}


class SimSettings(C.Structure):
    _fields_ = [
        ("has_eq_bands_v2", C.c_int32),
        ("eq_bands_v2", EqBandConfig * 10),
        ("deesser_enabled", C.c_int32),
        ("deesser_auto_enabled", C.c_int32),
        ("deesser_auto_amount", C.c_double),
        ("deesser_low_cut_hz", C.c_double),
        ("deesser_high_cut_hz", C.c_double),
        ("deesser_threshold_db", C.c_double),
        ("deesser_ratio", C.c_double),
        ("deesser_attack_ms", C.c_double),
        ("deesser_release_ms", C.c_double),
        ("deesser_max_reduction_db", C.c_double),
        ("eq_before_deesser", C.c_int32),
        ("compressor_enabled", C.c_int32),
        ("compressor_threshold_db", C.c_double),
        ("compressor_ratio", C.c_double),
        ("compressor_attack_ms", C.c_double),
        ("compressor_release_ms", C.c_double),
        ("compressor_makeup_gain_db", C.c_double),
        ("compressor_adaptive_release", C.c_int32),
        ("compressor_base_release_ms", C.c_double),
        ("compressor_auto_makeup_enabled", C.c_int32),
        ("compressor_target_lufs", C.c_double),
        ("compressor_sidechain_highpass_enabled", C.c_int32),
        ("limiter_enabled", C.c_int32),
        ("limiter_ceiling_db", C.c_double),
        ("limiter_careful_output_enabled", C.c_int32),
        ("limiter_lookahead_ms", C.c_double),
        ("limiter_release_ms", C.c_double),
    ]


class SimResult(C.Structure):
    _fields_ = [
        ("input_sample_peak_db", C.c_float),
        ("input_rms_db", C.c_float),
        ("output_sample_peak_db", C.c_float),
        ("pre_limiter_true_peak_db", C.c_float),
        ("output_true_peak_db", C.c_float),
        ("output_rms_db", C.c_float),
        ("limiter_effective_ceiling_db", C.c_float),
        ("sample_headroom_db", C.c_float),
        ("pre_limiter_true_peak_headroom_db", C.c_float),
        ("true_peak_headroom_db", C.c_float),
        ("limiter_gain_reduction_db", C.c_float),
        ("true_peak_limiter_gain_reduction_db", C.c_float),
        ("true_peak_limited_events", C.c_uint64),
        ("compressor_gain_reduction_db", C.c_float),
        ("deesser_gain_reduction_db", C.c_float),
        ("compressor_gain_reduction_median_db", C.c_float),
        ("compressor_gain_reduction_p95_db", C.c_float),
        ("compressor_gain_reduction_active_ratio", C.c_float),
        ("active_output_gain_db", C.c_float),
        ("silence_output_gain_db", C.c_float),
        ("silence_level_delta_db", C.c_float),
        ("compressor_pumping_score_db", C.c_float),
        ("non_finite_output", C.c_int32),
        ("deesser_gain_reduction_median_db", C.c_float),
        ("deesser_gain_reduction_p95_db", C.c_float),
        ("analysis_block_ms", C.c_float),
        ("active_analysis_threshold_db", C.c_float),
        ("active_analysis_block_count", C.c_uint64),
        ("processed_samples", C.c_uint64),
    ]


class EqV2Result(C.Structure):
    _fields_ = [
        ("input_sample_peak", C.c_float),
        ("output_sample_peak", C.c_float),
        ("input_true_peak", C.c_float),
        ("output_true_peak", C.c_float),
        ("input_rms", C.c_double),
        ("output_rms", C.c_double),
        ("max_response_db", C.c_double),
        ("sample_count", C.c_uint64),
        ("non_finite_output", C.c_int32),
    ]


class MakeupSettings(C.Structure):
    _fields_ = [
        ("threshold_db", C.c_double), ("ratio", C.c_double), ("attack_ms", C.c_double), ("release_ms", C.c_double),
        ("makeup_gain_db", C.c_double), ("target_lufs", C.c_double), ("vad_reliability", C.c_double),
        ("adaptive_release", C.c_int32), ("sidechain_highpass_enabled", C.c_int32),
    ]


MAKEUP_DEFAULTS = dict(threshold_db=-24.0, ratio=3.0, attack_ms=10.0, release_ms=180.0, makeup_gain_db=0.0,
                       target_lufs=-18.0, vad_reliability=1.0, adaptive_release=True, sidechain_highpass_enabled=True)
MAKEUP_TRACES = ("makeup_gain_db", "activity", "reliability", "gain_reduction_db", "input_rms_db", "output_rms_db")


class BlockStats(C.Structure):
    _fields_ = [
        ("input_sample_peak", C.c_float),
        ("output_sample_peak", C.c_float),
        ("true_peak_limiter_input_peak", C.c_float),
        ("output_true_peak", C.c_float),
        ("limiter_peak_gain_reduction_db", C.c_float),
        ("true_peak_limiter_gain_reduction_db", C.c_float),
        ("true_peak_limited_events", C.c_uint64),
        ("compressor_gain_reduction_db", C.c_float),
        ("deesser_gain_reduction_db", C.c_float),
    ]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(LIB_PATH))
        vp, d, i, f, sz = C.c_void_p, C.c_double, C.c_int, C.c_float, C.c_size_t
        fp = C.POINTER(C.c_float)
        dp = C.POINTER(C.c_double)
        L.afo_chain_new.restype = vp
        L.afo_chain_new.argtypes = [d]
        L.afo_chain_free.argtypes = [vp]
        for name in ("deesser", "eq", "compressor", "limiter", "tp_limiter"):
            fn = getattr(L, f"afo_chain_{name}")
            fn.restype = vp
            fn.argtypes = [vp]
        for name in ("deesser_enabled", "eq_enabled", "compressor_enabled", "limiter_enabled", "eq_before_deesser"):
            getattr(L, f"afo_chain_set_{name}").argtypes = [vp, i]
        L.afo_chain_process_block.restype = BlockStats
        L.afo_chain_process_block.argtypes = [vp, fp, sz]
        for name in ("gain", "frequency", "q"):
            getattr(L, f"afo_eq_set_band_{name}").argtypes = [vp, sz, d]
        L.afo_eq_set_band_config.argtypes = [vp, sz, C.POINTER(EqBandConfig)]
        L.afo_eq_reset.argtypes = [vp]
        for name in ("threshold", "ratio", "attack_time", "release_time", "base_release_time", "makeup_gain",
                     "target_lufs", "noise_reference_reliability", "limiter_feedback_gain_reduction_db"):
            getattr(L, f"afo_compressor_set_{name}").argtypes = [vp, d]
        for name in ("adaptive_release", "enabled", "auto_makeup_enabled", "sidechain_highpass_enabled"):
            getattr(L, f"afo_compressor_set_{name}").argtypes = [vp, i]
        for name in ("ceiling", "release_time", "lookahead_ms"):
            getattr(L, f"afo_limiter_set_{name}").argtypes = [vp, d]
        L.afo_limiter_set_enabled.argtypes = [vp, i]
        L.afo_tp_limiter_set_release_ms.argtypes = [vp, f]
        L.afo_tp_limiter_set_ceiling_linear.argtypes = [vp, f]
        for name in ("auto_amount", "low_cut_hz", "high_cut_hz", "threshold_db", "ratio", "attack_ms", "release_ms",
                     "max_reduction_db"):
            getattr(L, f"afo_deesser_set_{name}").argtypes = [vp, d]
        for name in ("enabled", "auto_enabled"):
            getattr(L, f"afo_deesser_set_{name}").argtypes = [vp, i]
        L.afo_sim_settings_default.argtypes = [C.POINTER(SimSettings)]
        L.afo_simulate_auto_eq_chain.restype = i
        L.afo_simulate_auto_eq_chain.argtypes = [fp, sz, d, dp, C.POINTER(SimSettings), C.POINTER(SimResult), fp]
        L.afo_simulate_eq_v2.restype = i
        L.afo_simulate_eq_v2.argtypes = [fp, sz, d, C.POINTER(EqBandConfig), C.POINTER(EqV2Result), fp]
        L.afo_eq_magnitude_response.restype = i
        L.afo_eq_magnitude_response.argtypes = [dp, sz, dp, d, dp]
        L.afo_eq_magnitude_response_v2.restype = i
        L.afo_eq_magnitude_response_v2.argtypes = [dp, sz, C.POINTER(EqBandConfig), d, dp]
        L.afo_kat_signal.argtypes = [fp, sz, C.c_uint64, d, d]
        L.afo_simulate_auto_makeup_control.restype = i
        L.afo_simulate_auto_makeup_control.argtypes = [fp, sz, d, dp, sz, d, d, C.POINTER(MakeupSettings), fp, fp]
        L.afo_measure_integrated_loudness.restype = i
        L.afo_measure_integrated_loudness.argtypes = [fp, sz, C.c_uint32, dp]
        L.afo_rnnoise_benchmark_frames.argtypes = [fp, fp, sz, C.c_uint64]
        L.afo_suppressor_init.argtypes = [vp, f, C.c_uint64]
        L.afo_suppressor_process.restype = sz
        L.afo_suppressor_process.argtypes = [vp, fp, fp, sz]
        L.afo_time_constant_to_coeff.restype = d
        L.afo_time_constant_to_coeff.argtypes = [d, d]
        L.afo_db_to_linear.restype = d
        L.afo_db_to_linear.argtypes = [d]
        L.afo_linear_to_db.restype = d
        L.afo_linear_to_db.argtypes = [d, d]
        L.afo_percentile_f32.restype = f
        L.afo_percentile_f32.argtypes = [fp, sz, f]
        L.afo_pumping_score.restype = f
        L.afo_pumping_score.argtypes = [fp, sz, f]
        _lib = L
    return _lib


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def kat_signal(n_blocks: int, noise_state: int = 0x6A09E667F3BCC909, f0: float = 180.0, phrase_hz: float = 1.7) -> np.ndarray:
    out = np.zeros(n_blocks * 480, dtype=np.float32)
    lib().afo_kat_signal(_fptr(out), n_blocks, C.c_uint64(noise_state & (2**64 - 1)), f0, phrase_hz)
    return out


def bands_v2_array(bands) -> "C.Array":
    arr = (EqBandConfig * 10)()
    for k, (name, freq, gain, q, slope, enabled) in enumerate(bands):
        if name not in EQ_TYPE_IDS:
            raise ValueError(f"band {k} has unsupported EQ filter type: {name}")
        arr[k] = EqBandConfig(EQ_TYPE_IDS[name], float(freq), float(gain), float(q), int(slope), int(bool(enabled)))
    return arr


def settings_from_dict(settings: dict | None) -> SimSettings:
    s = SimSettings()
    lib().afo_sim_settings_default(C.byref(s))
    for key, value in (settings or {}).items():
        if key == "eq_bands_v2":
            s.has_eq_bands_v2 = 1
            arr = bands_v2_array(value)
            for k in range(10):
                s.eq_bands_v2[k] = arr[k]
        elif key == "return_output_audio":
            continue
        elif hasattr(s, key):
            setattr(s, key, value)
        else:
            raise KeyError(key)
    return s


def simulate_auto_eq_chain(audio: np.ndarray, sample_rate: float, bands, settings: dict | None = None) -> dict:
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    if len(bands) != 10:
        raise ValueError(f"expected 10 EQ bands, got {len(bands)}")
    b = np.ascontiguousarray(np.asarray(bands, dtype=np.float64).reshape(10, 3))
    s = settings_from_dict(settings)
    r = SimResult()
    out = np.zeros_like(audio)
    rc = lib().afo_simulate_auto_eq_chain(_fptr(audio), audio.size, float(sample_rate), _dptr(b), C.byref(s), C.byref(r), _fptr(out))
    if rc != 0:
        raise ValueError("sample_rate must be positive and finite")
    d = {name: getattr(r, name) for name, _ in SimResult._fields_}
    d["non_finite_output"] = bool(d["non_finite_output"])
    if (settings or {}).get("return_output_audio", False):
        d["output_audio"] = out
    return d


def simulate_eq_v2(audio: np.ndarray, sample_rate: float, bands, return_output_audio: bool = False) -> dict:
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    if len(bands) != 10:
        raise ValueError(f"expected 10 EQ bands, got {len(bands)}")
    arr = bands_v2_array(bands)
    r = EqV2Result()
    out = np.zeros_like(audio)
    rc = lib().afo_simulate_eq_v2(_fptr(audio), audio.size, float(sample_rate), arr, C.byref(r), _fptr(out))
    if rc == -3:
        raise ValueError("audio must contain only finite samples")
    if rc != 0:
        raise ValueError("invalid EQ configuration or sample_rate")
    d = {name: getattr(r, name) for name, _ in EqV2Result._fields_}
    d["non_finite_output"] = bool(d["non_finite_output"])
    d["algorithmic_latency_samples"] = 0
    if return_output_audio:
        d["output_audio"] = out
    return d


def eq_magnitude_response(freqs, bands, sample_rate: float) -> np.ndarray:
    f = np.ascontiguousarray(freqs, dtype=np.float64)
    if len(bands) != 10:
        raise ValueError(f"expected 10 EQ bands, got {len(bands)}")
    b = np.ascontiguousarray(np.asarray(bands, dtype=np.float64).reshape(10, 3))
    out = np.zeros_like(f)
    rc = lib().afo_eq_magnitude_response(_dptr(f), f.size, _dptr(b), float(sample_rate), _dptr(out))
    if rc != 0:
        raise ValueError({-1: "sample_rate must be finite and positive", -2: "band frequency must be between 0 Hz and Nyquist", -3: "response frequencies must be finite and between 0 Hz and Nyquist"}[rc])
    return out


def eq_magnitude_response_v2(freqs, bands, sample_rate: float) -> np.ndarray:
    f = np.ascontiguousarray(freqs, dtype=np.float64)
    arr = bands_v2_array(bands)
    out = np.zeros_like(f)
    rc = lib().afo_eq_magnitude_response_v2(_dptr(f), f.size, arr, float(sample_rate), _dptr(out))
    if rc != 0:
        raise ValueError("invalid EQ v2 request")
    return out


class Chain:
    """Thin handle on afo_chain (OfflineDspBlockProcessor restatement)."""

    def __init__(self, sample_rate: float = 48000.0):
        self.L = lib()
        self.h = self.L.afo_chain_new(float(sample_rate))
        self.eq = self.L.afo_chain_eq(self.h)
        self.compressor = self.L.afo_chain_compressor(self.h)
        self.limiter = self.L.afo_chain_limiter(self.h)
        self.tp_limiter = self.L.afo_chain_tp_limiter(self.h)
        self.deesser = self.L.afo_chain_deesser(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.afo_chain_free(self.h)
            self.h = None

    def set(self, what: str, value) -> None:
        getattr(self.L, f"afo_chain_set_{what}")(self.h, int(value))

    def process_block(self, block: np.ndarray) -> BlockStats:
        assert block.dtype == np.float32 and block.flags.c_contiguous
        return self.L.afo_chain_process_block(self.h, _fptr(block), block.size)


def simulate_auto_makeup_control(audio, sample_rate, vad_probabilities, noise_floor_db, noise_reliability, settings=None) -> dict:
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    cfg = dict(MAKEUP_DEFAULTS)
    want_audio = False
    for k, v in (settings or {}).items():
        if k == "return_output_audio":
            want_audio = bool(v)
        else:
            cfg[k] = v
    s = MakeupSettings(**{k: (int(v) if isinstance(v, bool) else v) for k, v in cfg.items()})
    vad = np.ascontiguousarray(vad_probabilities, dtype=np.float64)
    blocks = (audio.size + 479) // 480
    traces = np.zeros((6, blocks), dtype=np.float32)
    out = np.zeros_like(audio)
    rc = lib().afo_simulate_auto_makeup_control(_fptr(audio), audio.size, float(sample_rate), _dptr(vad), vad.size,
                                                float(noise_floor_db), float(noise_reliability), C.byref(s),
                                                traces.ctypes.data_as(C.POINTER(C.c_float)), _fptr(out))
    if rc != 0:
        raise ValueError(f"simulate_auto_makeup_control rejected its arguments ({rc})")
    d = {name: traces[k].copy() for k, name in enumerate(MAKEUP_TRACES)}
    d.update(control_block_size=480, control_cadence_hz=sample_rate / 480.0, processed_samples=int(audio.size))
    if want_audio:
        d["output_audio"] = out
    return d


def measure_integrated_loudness(audio, sample_rate: int) -> float:
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    v = C.c_double(0.0)
    rc = lib().afo_measure_integrated_loudness(_fptr(audio), audio.size, int(sample_rate), C.byref(v))
    if rc != 0:
        raise ValueError(f"integrated loudness unavailable ({rc})")
    return v.value


def rnnoise_benchmark_frames(audio, weight_seed: int = 0x5EED) -> np.ndarray:
    """bin/rnnoise_benchmark.rs:51-117 protocol over the RNNoise restatement (parity unpinned)."""
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    out = np.zeros_like(audio)
    lib().afo_rnnoise_benchmark_frames(_fptr(audio), _fptr(out), audio.size, C.c_uint64(weight_seed))
    return out


def suppressor_process(audio, strength: float = 1.0, weight_seed: int = 0x5EED) -> np.ndarray:
    """RNNoiseProcessor (rnnoise.rs:122-164) over whole frames; returns len(audio)//480*480 samples."""
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    state = C.create_string_buffer(1 << 18)  # afo_suppressor is ~140 KB
    lib().afo_suppressor_init(state, float(strength), C.c_uint64(weight_seed))
    out = np.zeros_like(audio)
    n = lib().afo_suppressor_process(state, _fptr(out), _fptr(audio), audio.size)
    return out[:n]


def set_rnn_eval_order(order: int) -> None:
    """0 = wavefront-native sums (what the GPU evaluates), 1 = the published scalar C's running sums (af_rnnoise.c)."""
    C.c_int.in_dll(lib(), "afo_rnn_eval_order").value = int(order)


def suppressor_process_traced(audio, strength: float = 1.0, weight_seed: int = 0x5EED, eval_order: int = 0):
    """suppressor_process plus the per-frame (pitch index, silence flag) decisions; `eval_order` as set_rnn_eval_order."""
    audio = np.ascontiguousarray(audio, dtype=np.float32)
    L = lib()
    L.afo_suppressor_process_traced.restype = C.c_size_t
    L.afo_suppressor_process_traced.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t,
                                                C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    state = C.create_string_buffer(1 << 18)
    L.afo_suppressor_init(state, float(strength), C.c_uint64(weight_seed))
    frames = audio.size // 480
    out = np.zeros_like(audio)
    pitch = np.zeros(frames, dtype=np.int32)
    silence = np.zeros(frames, dtype=np.int32)
    set_rnn_eval_order(eval_order)
    try:
        n = L.afo_suppressor_process_traced(state, _fptr(out), _fptr(audio), audio.size,
                                            pitch.ctypes.data_as(C.POINTER(C.c_int32)), silence.ctypes.data_as(C.POINTER(C.c_int32)))
    finally:
        set_rnn_eval_order(0)
    return out[:n], pitch, silence


def scale_sample_for_model(sample: float) -> float:
    """RNNoiseProcessor::scale_sample_for_model (rnnoise.rs:89-111)."""
    L = lib()
    L.afo_scale_sample_for_model.restype = C.c_float
    L.afo_scale_sample_for_model.argtypes = [C.c_float]
    return float(L.afo_scale_sample_for_model(C.c_float(sample)))


class RNNoiseProcessor:
    """The reference's RNNoiseProcessor surface (rnnoise.rs:25-245) over the restatement, rings included."""

    def __init__(self, strength: float = 1.0, weight_seed: int = 0x5EED):
        L = lib()
        sz, vp, fp = C.c_size_t, C.c_void_p, C.POINTER(C.c_float)
        L.afo_processor_init.argtypes = [vp, C.c_float, C.c_uint64]
        L.afo_processor_set_strength.argtypes = [vp, C.c_float]
        L.afo_processor_get_strength.restype = C.c_float
        L.afo_processor_get_strength.argtypes = [vp]
        L.afo_processor_push_samples.restype = sz
        L.afo_processor_push_samples.argtypes = [vp, fp, sz]
        L.afo_processor_process_frames.argtypes = [vp]
        for name in ("available_samples", "pending_input"):
            fn = getattr(L, f"afo_processor_{name}")
            fn.restype = sz
            fn.argtypes = [vp]
        for name in ("read_samples", "drain_pending_input"):
            fn = getattr(L, f"afo_processor_{name}")
            fn.restype = sz
            fn.argtypes = [vp, fp, sz]
        L.afo_processor_set_enabled.argtypes = [vp, C.c_int]
        L.afo_processor_soft_reset.argtypes = [vp]
        L.afo_processor_reset.argtypes = [vp]
        self._L = L
        self._p = C.create_string_buffer(1 << 18)  # afo_rnnoise_processor is ~210 KB
        L.afo_processor_init(self._p, float(strength), C.c_uint64(weight_seed))

    def set_strength(self, v: float) -> None:
        self._L.afo_processor_set_strength(self._p, float(v))

    def get_strength(self) -> float:
        return float(self._L.afo_processor_get_strength(self._p))

    def push_samples(self, samples) -> int:
        a = np.ascontiguousarray(samples, dtype=np.float32)
        return int(self._L.afo_processor_push_samples(self._p, _fptr(a), a.size))

    def process_frames(self) -> None:
        self._L.afo_processor_process_frames(self._p)

    def available_samples(self) -> int:
        return int(self._L.afo_processor_available_samples(self._p))

    def pending_input(self) -> int:
        return int(self._L.afo_processor_pending_input(self._p))

    def read_samples(self, count: int) -> np.ndarray:
        out = np.zeros(int(count), dtype=np.float32)
        n = int(self._L.afo_processor_read_samples(self._p, _fptr(out), out.size))
        return out[:n]

    def drain_pending_input(self) -> np.ndarray:
        out = np.zeros(8192 + 480, dtype=np.float32)
        n = int(self._L.afo_processor_drain_pending_input(self._p, _fptr(out), out.size))
        return out[:n]

    def set_enabled(self, on: bool) -> None:
        self._L.afo_processor_set_enabled(self._p, int(bool(on)))

    def soft_reset(self) -> None:
        self._L.afo_processor_soft_reset(self._p)

    def reset(self) -> None:
        self._L.afo_processor_reset(self._p)


def prefilter(audio, sample_rate: float = 48000.0) -> np.ndarray:
    """DC block + 80 Hz high-pass (routing.rs:826-843)."""
    class Pre(C.Structure):
        _fields_ = [("dc_x1", C.c_float), ("dc_y1", C.c_float), ("hp", C.c_byte * 256)]

    L = lib()
    L.afo_prefilter_init.argtypes = [C.c_void_p, C.c_double]
    L.afo_prefilter_process_block.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.c_int]
    pre = Pre()
    L.afo_prefilter_init(C.byref(pre), float(sample_rate))
    out = np.ascontiguousarray(audio, dtype=np.float32).copy()
    L.afo_prefilter_process_block(C.byref(pre), _fptr(out), out.size, 1)
    return out


RESAMPLER_WINDOWS = {"blackman_harris": 0, "blackman_harris_squared": 1, "blackman": 2, "blackman_squared": 3,
                     "hann": 4, "hann_squared": 5}


def simulate_product_resampler(samples, input_rate: int, output_rate: int, chunk_size: int = 1024,
                               sinc_len: int | None = None, window: str | None = None, f_cutoff: float = 0.0):
    """resampling.rs:179-261 over the rubato restatement (oracle/af_resampler.c; sample parity unpinned).
    Returns (output ndarray f64, delay, expected_frames, blocks)."""
    L = lib()
    dp = C.POINTER(C.c_double)
    L.afo_simulate_product_resampler.restype = C.c_int64
    L.afo_simulate_product_resampler.argtypes = [dp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_size_t, C.c_size_t, C.c_int,
                                                 C.c_float, dp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                                 C.POINTER(C.c_size_t)]
    x = np.ascontiguousarray(samples, dtype=np.float64)
    sinc_len = 128 if sinc_len is None else int(sinc_len)
    win = RESAMPLER_WINDOWS["blackman" if window is None else window]
    ratio = output_rate / input_rate
    cap = int((x.size + 4 * chunk_size + 2 * sinc_len) * ratio * 1.01) + 4096
    out = np.zeros(cap, dtype=np.float64)
    delay, expected, blocks = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    n = L.afo_simulate_product_resampler(x.ctypes.data_as(dp), x.size, int(input_rate), int(output_rate), int(chunk_size),
                                         sinc_len, win, float(f_cutoff), out.ctypes.data_as(dp), cap, C.byref(delay),
                                         C.byref(expected), C.byref(blocks))
    if n < 0:
        raise RuntimeError(f"output capacity too small ({-n} frames needed)")
    return out[:n], int(delay.value), int(expected.value), int(blocks.value)


def resampler_calculate_cutoff(sinc_len: int = 128, window: str = "blackman") -> float:
    L = lib()
    L.afo_resampler_calculate_cutoff.restype = C.c_float
    L.afo_resampler_calculate_cutoff.argtypes = [C.c_size_t, C.c_int]
    return float(L.afo_resampler_calculate_cutoff(int(sinc_len), RESAMPLER_WINDOWS[window]))


class Gate:
    """dsp/gate.rs NoiseGate without a VadAutoGate attached (the expander path)."""

    class _State(C.Structure):
        _fields_ = [("threshold_db", C.c_double), ("attack_coeff", C.c_double), ("release_coeff", C.c_double),
                    ("rms_coeff", C.c_double), ("sample_rate", C.c_double), ("rms_envelope_sq", C.c_double),
                    ("detector_level_db", C.c_double), ("current_gain", C.c_double), ("hold_remaining_samples", C.c_size_t),
                    ("is_open", C.c_int), ("enabled", C.c_int), ("vad_mode", C.c_int), ("effective_gate_open", C.c_int),
                    ("has_effective_gate_state", C.c_int), ("chatter_window_remaining_samples", C.c_size_t),
                    ("chatter_cooldown_samples", C.c_size_t), ("auto_relax_remaining_samples", C.c_size_t),
                    ("chatter_transition_count", C.c_uint32), ("chatter_event_count", C.c_uint64)]

    def __init__(self, threshold_db=-40.0, attack_ms=10.0, release_ms=100.0, sample_rate=48_000.0, vad_mode=False):
        L = lib()
        L.afo_gate_init.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
        L.afo_gate_set_vad_mode.argtypes = [C.c_void_p, C.c_int]
        L.afo_gate_process_block.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_size_t]
        self.s = Gate._State()
        L.afo_gate_init(C.byref(self.s), threshold_db, attack_ms, release_ms, sample_rate)
        L.afo_gate_set_vad_mode(C.byref(self.s), int(vad_mode))

    def process(self, audio: np.ndarray) -> np.ndarray:
        out = np.ascontiguousarray(audio, dtype=np.float32).copy()
        lib().afo_gate_process_block(C.byref(self.s), _fptr(out), out.size)
        return out

    @property
    def current_gain(self) -> float:
        return float(np.float32(self.s.current_gain))

    @property
    def is_open(self) -> bool:
        return bool(self.s.is_open)

    @property
    def chatter_event_count(self) -> int:
        return int(self.s.chatter_event_count)


def simulate_gate_suppressor_order(audio, vad_probabilities, suppressor_before_gate: bool, suppressor_strength: float = 1.0,
                                   settings: dict | None = None, weight_seed: int = 0x5EED) -> dict:
    """python_api.rs:288-376 over the restatements (480-sample frames, last one zero padded)."""
    settings = settings or {}
    x = np.ascontiguousarray(audio, dtype=np.float32)
    n = x.size
    frames = -(-n // 480)
    padded = np.zeros(frames * 480, dtype=np.float32)
    padded[:n] = x
    gate = Gate(settings.get("gate_threshold_db", -40.0), settings.get("gate_attack_ms", 10.0),
                settings.get("gate_release_ms", 100.0), 48_000.0, vad_mode=True)
    gains = []

    def run_gate(sig):
        out = np.empty_like(sig)
        for f in range(frames):
            out[f * 480 : (f + 1) * 480] = gate.process(sig[f * 480 : (f + 1) * 480])
            gains.append(gate.current_gain)
        return out

    if suppressor_before_gate:
        y = run_gate(suppressor_process(padded, suppressor_strength, weight_seed))
    else:
        y = suppressor_process(run_gate(padded), suppressor_strength, weight_seed)
    return {"output_audio": y[:n], "gate_gain": gains, "gate_chatter_event_count": gate.chatter_event_count,
            "gate_noise_floor_db": -60.0, "gate_noise_floor_reliability": 0.0, "suppressor_latency_samples": 480}
