"""audio-forge_amd: MI355X-native batched voice-DSP engine behind the mic_eq.mic_eq_core surface.

The directory name carries a hyphen (it is the package name the build contract asks for), so
import it with ``importlib.import_module("audio-forge_amd")`` or put this directory on
``sys.path`` and ``import mic_eq_mi`` -- both give the same module objects.
"""
import importlib
import pathlib
import sys

_HERE = pathlib.Path(__file__).resolve().parent
if str(_HERE) not in sys.path:
    sys.path.insert(0, str(_HERE))

mic_eq_mi = importlib.import_module("mic_eq_mi")
from mic_eq_mi import *  # noqa: F401,F403,E402
