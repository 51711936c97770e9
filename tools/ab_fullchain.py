"""Same-box A/B of execution options of the full chain (prefilter + suppressor + chain): writes the output and the block rows
for a seeded batch to gpurun_out/abfc_<tag>.npz (run once per variant, e.g. AF_EQ_OFFLOAD=0 / =1) or compares two runs bit
for bit (`python tools/ab_fullchain.py cmp a b`)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "audio-forge_amd"))

if sys.argv[1] == "cmp":
    a = np.load(os.path.join(ROOT, "gpurun_out", f"abfc_{sys.argv[2]}.npz"))
    b = np.load(os.path.join(ROOT, "gpurun_out", f"abfc_{sys.argv[3]}.npz"))
    same = np.array_equal(a["y"].view(np.uint32), b["y"].view(np.uint32)) and a["rows"].tobytes() == b["rows"].tobytes()
    print("bit-identical (audio and block rows)" if same else f"DIFFERENT: audio max abs {np.abs(a['y'] - b['y']).max():.3e}")
    sys.exit(0 if same else 1)

import signals as S
from mic_eq_mi import mic_eq_core as core

audio = S.batch_signal(70, 230)  # 70 streams (64 + 6), 2.3 s: ramped windows, two calls
bands = [(80.0 * 1.75**i, 3.0 if i % 2 else -2.5, 1.0) for i in range(10)]  # legacy setters: a crossfade opens the stream
eng = core.Engine(48_000.0, 70)
# AB_MODE: "full" (default: front end + suppressor + chain), "dynamics" (no suppressor), "+automakeup" appended: the compressor's
# auto-makeup on -- the execution forms differ per mode (one chain launch following the suppressor / following the systolic EQ)
mode = os.environ.get("AB_MODE", "full")
settings = dict(S.limiter_settings(2.0))
if "+automakeup" in mode:
    settings.update(compressor_auto_makeup_enabled=True, compressor_target_lufs=-18.0)
if "+steep" in mode:  # typed bands with steep slopes: 15 EQ sections instead of 10 (odd: the two-wave EQ kernel splits them 7 + 8)
    steep = list(S.DEFAULT_TYPED_BANDS)
    steep[0] = ("high_pass", 90.0, 0.0, 0.707, 48, True)    # four sections
    steep[9] = ("low_pass", 15000.0, 0.0, 0.707, 36, True)  # three sections
    steep[4] = ("bell", 1000.0, 6.0, 2.0, 12, True)
    settings["eq_bands_v2"] = steep
core.configure_auto_eq_chain(eng, 48_000.0, bands, settings)
if mode.startswith("full"):
    eng.set_prefilter_enabled(1, 1)
    eng.set_suppressor_enabled(1)
ys, rows = [], []
for lo, hi in ((0, 130 * 480), (130 * 480, 230 * 480)):
    ys.append(eng.process(audio[:, lo:hi]))
    rows.append(eng.block_stats().copy())
eng.close()
np.savez(os.path.join(ROOT, "gpurun_out", f"abfc_{sys.argv[1]}.npz"), y=np.concatenate(ys, axis=1), rows=np.concatenate(rows, axis=0))
print(sys.argv[1], float(np.sqrt(np.mean(np.concatenate(ys, axis=1).astype(np.float64) ** 2))))
