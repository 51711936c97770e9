"""mic_eq_mi -- the MI355X backend behind the reference's ``mic_eq`` offline operator names.

Mirrors python/mic_eq/__init__.py:38-98 of the reference: the names below are the ones the
evaluation harness imports from ``mic_eq``; an operator this backend does not build yet raises
ImportError through ``_missing_core`` (the reference's own degradation path), and nothing here
falls back to a CPU implementation.
"""
from __future__ import annotations

from . import _lib
from ._lib import KERNEL_AUTO, KERNEL_LANE_PER_STREAM, KERNEL_PHASED, KERNEL_QUAD, KERNEL_STAGED, LAYOUT_STREAM_MAJOR, LAYOUT_TIME_MAJOR

_CORE_IMPORT_ERROR = None
try:
    from . import mic_eq_core as _core_module
    _lib.load()
except ImportError as error:  # the HIP library is missing or unloadable
    _core_module = None
    _CORE_IMPORT_ERROR = error


def _missing_core(*args, **kwargs):
    raise ImportError(
        "mic_eq_mi backend (libaudioforge_mi.so) is unavailable or missing this API. Build it with: "
        "python -c 'import __graft_entry__ as g; g.build()'"
    ) from _CORE_IMPORT_ERROR


_OPERATORS = (
    "simulate_auto_eq_chain",
    "simulate_auto_eq_chain_batch",
    "simulate_auto_makeup_control",
    "simulate_gate_suppressor_order",
    "simulate_eq_v2",
    "simulate_product_resampler",
    "product_resampler_configuration",
    "eq_magnitude_response",
    "eq_magnitude_response_v2",
    "measure_integrated_loudness",
    "suppress",
    "rnnoise_benchmark",
)

CORE_AVAILABLE = _core_module is not None
for _name in _OPERATORS:
    globals()[_name] = getattr(_core_module, _name, _missing_core) if _core_module is not None else _missing_core
Engine = getattr(_core_module, "Engine", None) if _core_module is not None else None
NoiseSuppressor = getattr(_core_module, "NoiseSuppressor", None) if _core_module is not None else None
NoiseModel = getattr(_core_module, "NoiseModel", None) if _core_module is not None else None
new_noise_suppression_engine = (getattr(_core_module, "new_noise_suppression_engine", _missing_core)
                                if _core_module is not None else _missing_core)

__all__ = ["CORE_AVAILABLE", "Engine", "NoiseSuppressor", "NoiseModel", "new_noise_suppression_engine", *_OPERATORS, "LAYOUT_STREAM_MAJOR", "LAYOUT_TIME_MAJOR", "KERNEL_AUTO",
           "KERNEL_LANE_PER_STREAM", "KERNEL_PHASED", "KERNEL_QUAD", "KERNEL_STAGED"]
