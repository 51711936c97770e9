"""Loader for libaudioforge_mi.so (the C ABI of include/audioforge_mi.h).

There is no CPU fallback: if the HIP library is missing or does not load, every call
raises ImportError, exactly like the reference's `_missing_core`
(python/mic_eq/__init__.py:48-52).
"""
from __future__ import annotations

import ctypes as C
import pathlib

PKG_DIR = pathlib.Path(__file__).resolve().parents[1]
import os

# AF_LIB_PATH: load another build of the same library (same-box A/B runs of kernel variants)
LIB_PATH = pathlib.Path(os.environ.get("AF_LIB_PATH") or (PKG_DIR / "libaudioforge_mi.so"))

AF_OK = 0
AF_ERR_INVALID_ARGUMENT = -1
AF_ERR_BACKEND = -2
AF_ERR_NON_FINITE = -3
AF_ERR_STATE = -4
AF_ERR_UNSUPPORTED = -5

LAYOUT_STREAM_MAJOR = 0
LAYOUT_TIME_MAJOR = 1
KERNEL_AUTO, KERNEL_LANE_PER_STREAM, KERNEL_PHASED, KERNEL_QUAD, KERNEL_STAGED, KERNEL_ROLES = 0, 1, 2, 3, 4, 5


class EqBandConfig(C.Structure):
    _fields_ = [
        ("filter_type", C.c_int32),
        ("frequency_hz", C.c_double),
        ("gain_db", C.c_double),
        ("q", C.c_double),
        ("slope_db_per_octave", C.c_int32),
        ("enabled", C.c_int32),
    ]


class BlockStats(C.Structure):
    _fields_ = [
        ("input_sample_peak", C.c_float),
        ("output_sample_peak", C.c_float),
        ("true_peak_limiter_input_peak", C.c_float),
        ("output_true_peak", C.c_float),
        ("limiter_peak_gain_reduction_db", C.c_float),
        ("true_peak_limiter_gain_reduction_db", C.c_float),
        ("compressor_gain_reduction_db", C.c_float),
        ("deesser_gain_reduction_db", C.c_float),
        ("input_square_sum", C.c_double),
        ("output_square_sum", C.c_double),
        ("true_peak_limited_events", C.c_uint32),
        ("non_finite_output", C.c_uint32),
        ("compressor_makeup_gain_db", C.c_float),
        ("auto_makeup_activity", C.c_float),
        ("auto_makeup_reliability", C.c_float),
        ("reserved", C.c_float),
    ]


# every symbol include/audioforge_mi.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _d, _f, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_float, C.c_size_t
_fp, _dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
SIGNATURES = {
    "af_version": (C.c_int, []),
    "af_last_error": (C.c_char_p, []),
    "af_device_count": (C.c_int, []),
    "af_engine_create": (C.c_int, [_d, _i32, _i32, C.POINTER(_vp)]),
    "af_engine_destroy": (None, [_vp]),
    "af_engine_reset": (C.c_int, [_vp]),
    "af_engine_n_streams": (_i32, [_vp]),
    "af_engine_set_deesser_enabled": (C.c_int, [_vp, _i32]),
    "af_engine_set_eq_enabled": (C.c_int, [_vp, _i32]),
    "af_engine_set_compressor_enabled": (C.c_int, [_vp, _i32]),
    "af_engine_set_limiter_enabled": (C.c_int, [_vp, _i32]),
    "af_engine_set_eq_before_deesser": (C.c_int, [_vp, _i32]),
    "af_engine_set_control_block_samples": (C.c_int, [_vp, _i32]),
    "af_engine_set_input_scrub_enabled": (C.c_int, [_vp, _i32]),
    "af_engine_set_input_clamp_enabled": (C.c_int, [_vp, _i32]),
    "af_engine_set_prefilter_enabled": (C.c_int, [_vp, _i32, _i32]),
    "af_engine_set_suppressor_enabled": (C.c_int, [_vp, _i32]),
    "af_engine_set_suppressor_strength": (C.c_int, [_vp, _f]),
    "af_suppressor_set_synthetic_weights": (C.c_int, [_vp, C.c_uint64]),
    "af_suppressor_load_weights": (C.c_int, [_vp, C.c_char_p, _sz]),
    "af_suppressor_set_raw_protocol": (C.c_int, [_vp, _i32]),
    "af_suppressor_latency_samples": (_i32, [_vp]),
    "af_suppressor_debug_read": (C.c_int, [_vp, _i32, _i32, _fp, _fp, _fp]),
    "af_eq_set_band_frequency": (C.c_int, [_vp, _i32, _d]),
    "af_eq_set_band_gain": (C.c_int, [_vp, _i32, _d]),
    "af_eq_set_band_q": (C.c_int, [_vp, _i32, _d]),
    "af_eq_set_band_config": (C.c_int, [_vp, _i32, C.POINTER(EqBandConfig)]),
    "af_eq_reset": (C.c_int, [_vp]),
    "af_eq_band_config_validate": (C.c_int, [C.POINTER(EqBandConfig), _i32, _d]),
    "af_compressor_set_threshold": (C.c_int, [_vp, _d]),
    "af_compressor_set_ratio": (C.c_int, [_vp, _d]),
    "af_compressor_set_attack_time": (C.c_int, [_vp, _d]),
    "af_compressor_set_release_time": (C.c_int, [_vp, _d]),
    "af_compressor_set_makeup_gain": (C.c_int, [_vp, _d]),
    "af_compressor_set_adaptive_release": (C.c_int, [_vp, _i32]),
    "af_compressor_set_base_release_time": (C.c_int, [_vp, _d]),
    "af_compressor_set_auto_makeup_enabled": (C.c_int, [_vp, _i32]),
    "af_compressor_set_target_lufs": (C.c_int, [_vp, _d]),
    "af_compressor_set_sidechain_highpass_enabled": (C.c_int, [_vp, _i32]),
    "af_compressor_set_noise_reference_reliability": (C.c_int, [_vp, _d]),
    "af_compressor_set_activity_evidence": (C.c_int, [_vp, _dp, _i64, _i32, _d, _d, _d]),
    "af_limiter_set_ceiling": (C.c_int, [_vp, _d]),
    "af_limiter_set_release_time": (C.c_int, [_vp, _d]),
    "af_limiter_set_lookahead_ms": (C.c_int, [_vp, _d]),
    "af_limiter_ceiling_db": (_d, [_vp]),
    "af_limiter_lookahead_samples": (_i32, [_vp]),
    "af_true_peak_limiter_set_release_ms": (C.c_int, [_vp, _f]),
    "af_deesser_set_auto_enabled": (C.c_int, [_vp, _i32]),
    "af_deesser_set_auto_amount": (C.c_int, [_vp, _d]),
    "af_deesser_set_low_cut_hz": (C.c_int, [_vp, _d]),
    "af_deesser_set_high_cut_hz": (C.c_int, [_vp, _d]),
    "af_deesser_set_threshold_db": (C.c_int, [_vp, _d]),
    "af_deesser_set_ratio": (C.c_int, [_vp, _d]),
    "af_deesser_set_attack_ms": (C.c_int, [_vp, _d]),
    "af_deesser_set_release_ms": (C.c_int, [_vp, _d]),
    "af_deesser_set_max_reduction_db": (C.c_int, [_vp, _d]),
    "af_engine_process_device": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "af_engine_process_host": (C.c_int, [_vp, _fp, _fp, _i64, _i32]),
    "af_engine_synchronize": (C.c_int, [_vp]),
    "af_engine_stream_host": (C.c_int, [_vp, _fp, _i64, _fp, _i64, C.POINTER(_i64)]),
    "af_engine_pending_input": (_i64, [_vp]),
    "af_engine_last_output_samples": (_i64, [_vp]),
    "af_suppressor_debug_scale_for_model": (C.c_int, [_fp, _fp, _i64, _i32]),
    "af_suppressor_set_trace_enabled": (C.c_int, [_vp, _i32]),
    "af_suppressor_trace_frames": (_i64, [_vp]),
    "af_suppressor_read_trace": (C.c_int, [_vp, C.POINTER(_i32), _i64]),
    # NoiseSuppressor (noise_suppressor.rs:18-194)
    "af_noise_model_from_id": (C.c_int, [C.c_char_p, C.POINTER(_i32)]),
    "af_noise_model_id": (C.c_char_p, [_i32]),
    "af_noise_model_display_name": (C.c_char_p, [_i32]),
    "af_noise_model_available": (_i32, [C.POINTER(_i32), _i32]),
    "af_noise_suppressor_create": (C.c_int, [_i32, _i32, _i32, C.POINTER(_vp)]),
    "af_noise_suppressor_destroy": (None, [_vp]),
    "af_noise_suppressor_engine": (_vp, [_vp]),
    "af_noise_suppressor_push_samples": (_i64, [_vp, _fp, _i64, _i64]),
    "af_noise_suppressor_process_frames": (C.c_int, [_vp]),
    "af_noise_suppressor_available_samples": (_i64, [_vp]),
    "af_noise_suppressor_pending_input": (_i64, [_vp]),
    "af_noise_suppressor_pop_samples_into": (_i64, [_vp, _fp, _i64, _i64]),
    "af_noise_suppressor_drain_pending_input": (_i64, [_vp, _fp, _i64, _i64]),
    "af_noise_suppressor_set_strength": (C.c_int, [_vp, _f]),
    "af_noise_suppressor_get_strength": (_f, [_vp]),
    "af_noise_suppressor_set_enabled": (C.c_int, [_vp, _i32]),
    "af_noise_suppressor_is_enabled": (_i32, [_vp]),
    "af_noise_suppressor_soft_reset": (C.c_int, [_vp]),
    "af_noise_suppressor_reset": (C.c_int, [_vp]),
    "af_noise_suppressor_model_type": (_i32, [_vp]),
    "af_noise_suppressor_latency_samples": (_i32, [_vp]),
    "af_noise_suppressor_backend_available": (_i32, [_vp]),
    "af_noise_suppressor_backend_failed": (_i32, [_vp]),
    "af_noise_suppressor_backend_error": (C.c_char_p, [_vp]),
    "af_engine_last_block_count": (_i64, [_vp]),
    "af_engine_read_block_stats": (C.c_int, [_vp, C.POINTER(BlockStats), _i64]),
    "af_engine_samples_processed": (_i64, [_vp]),
    "af_engine_set_preset_count": (C.c_int, [_vp, _i32]),
    "af_engine_preset_count": (_i32, [_vp]),
    "af_engine_select_preset": (C.c_int, [_vp, _i32]),
    "af_engine_assign_presets": (C.c_int, [_vp, C.POINTER(_i32), _i32]),
    "af_engine_set_kernel": (C.c_int, [_vp, _i32]),
    "af_engine_set_ring_variant": (C.c_int, [_vp, _i32, _i32]),
    "af_engine_set_timing_enabled": (C.c_int, [_vp, _i32]),
    "af_engine_last_kernel_ms": (C.c_int, [_vp, _dp, C.POINTER(_i32)]),
    "af_engine_last_stage_ms": (C.c_int, [_vp, _dp, _dp]),
    "af_engine_last_chain_launch_ms": (C.c_int, [_vp, _dp, _dp, C.POINTER(_i32)]),
    # product resampler
    "af_engine_last_kernel": (C.c_int, [_vp]),
    "af_resampler_calculate_cutoff": (C.c_int, [_i32, _i32, C.POINTER(C.c_float)]),
    "af_resampler_create": (C.c_int, [C.c_uint32, C.c_uint32, _i64, _i32, _i32, _i32, C.POINTER(_vp)]),
    "af_resampler_destroy": (None, [_vp]),
    "af_resampler_output_delay": (C.c_int, [_vp]),
    "af_resampler_expected_frames": (_i64, [_vp, _i64]),
    "af_resampler_sinc_len": (C.c_int, [_vp]),
    "af_resampler_copy_sinc_table": (C.c_int, [_vp, _dp]),
    "af_resampler_plan": (C.c_int, [_vp, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "af_resampler_process_device": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i64, _i64, _vp]),
    "af_resampler_process_host": (C.c_int, [_vp, _dp, _dp, _i64, _i32, _i64, _i64]),
    "af_resampler_last_kernel_ms": (C.c_int, [_vp, _dp]),
    "af_gate_process_host": (C.c_int, [_fp, _fp, _i64, _i32, _i64, _d, _d, _d, _d, _i32, _i32, _fp, C.POINTER(C.c_uint64), _i32]),
    "af_measure_integrated_loudness_device": (C.c_int, [_vp, _i64, _i32, _i64, C.c_uint32, _i32, _dp, C.POINTER(_i32)]),
    "af_measure_integrated_loudness_host": (C.c_int, [_fp, _i64, _i32, _i64, C.c_uint32, _i32, _dp, C.POINTER(_i32)]),
    "af_eq_magnitude_response": (C.c_int, [_dp, _sz, _dp, _d, _dp]),
    "af_eq_magnitude_response_v2": (C.c_int, [_dp, _sz, C.POINTER(EqBandConfig), _d, _dp]),
    "af_engine_eq_magnitude_response": (C.c_int, [_vp, _dp, _sz, _dp]),
}

# entry points whose return value is data, not an af_status
VALUE_FUNCTIONS = {
    "af_version", "af_last_error", "af_device_count", "af_engine_n_streams", "af_limiter_ceiling_db",
    "af_limiter_lookahead_samples", "af_suppressor_latency_samples", "af_engine_last_block_count", "af_engine_samples_processed", "af_engine_destroy", "af_engine_last_kernel",
    "af_engine_pending_input", "af_engine_last_output_samples", "af_suppressor_trace_frames", "af_engine_preset_count",
    "af_noise_model_id", "af_noise_model_display_name", "af_noise_model_available", "af_noise_suppressor_destroy",
    "af_noise_suppressor_engine", "af_noise_suppressor_push_samples", "af_noise_suppressor_available_samples",
    "af_noise_suppressor_pending_input", "af_noise_suppressor_pop_samples_into", "af_noise_suppressor_drain_pending_input",
    "af_noise_suppressor_get_strength", "af_noise_suppressor_is_enabled", "af_noise_suppressor_model_type",
    "af_noise_suppressor_latency_samples", "af_noise_suppressor_backend_available", "af_noise_suppressor_backend_failed",
    "af_noise_suppressor_backend_error",
    "af_resampler_destroy", "af_resampler_output_delay", "af_resampler_expected_frames", "af_resampler_sinc_len",
}

_lib = None
_load_error: Exception | None = None


def load() -> C.CDLL:
    """Load the HIP library or raise ImportError (never falls back to a CPU path)."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise ImportError(str(_load_error)) from _load_error
    try:
        if not LIB_PATH.exists():
            raise OSError(f"{LIB_PATH} is missing; build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(str(LIB_PATH))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
    except (OSError, AttributeError) as error:
        _load_error = error
        raise ImportError(
            f"libaudioforge_mi.so (the MI355X HIP backend) is unavailable: {error}. There is no CPU fallback."
        ) from error
    _lib = lib
    return lib


def last_error() -> str:
    return load().af_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map af_status to the exception classes the reference binding raises."""
    if rc == AF_OK:
        return
    msg = last_error()
    if rc in (AF_ERR_INVALID_ARGUMENT, AF_ERR_NON_FINITE):
        raise ValueError(msg)  # PyValueError, lib.rs:105-141,225-229
    if rc == AF_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)  # PyRuntimeError, python_api.rs:332-340
