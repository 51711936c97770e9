"""Same-box A/B of suppressor kernel variants: writes the suppressor output for a seeded batch to
gpurun_out/ab_<tag>.npy (run once per variant, e.g. AF_RNN_VARIANT=1 / =4), or compares two such files
bit for bit (`python tools/ab_suppressor.py cmp a b`)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "audio-forge_amd"))

if sys.argv[1] == "cmp":
    a = np.load(os.path.join(ROOT, "gpurun_out", f"ab_{sys.argv[2]}.npy"))
    b = np.load(os.path.join(ROOT, "gpurun_out", f"ab_{sys.argv[3]}.npy"))
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    print("bit-identical" if same else f"DIFFERENT: max abs {np.abs(a - b).max():.3e}, {np.count_nonzero(a != b)} samples")
    sys.exit(0 if same else 1)

import signals as S
import mic_eq_mi

audio = S.batch_signal(52, 160)  # 52 streams: three full 16-stream groups + 4; 1.6 s = 3 windows + 10 frames
strength = float(os.environ.get("AB_STRENGTH", "1.0"))  # < 1: the wet/dry mix path
out = mic_eq_mi.suppress(audio, strength, 0x5EED)
np.save(os.path.join(ROOT, "gpurun_out", f"ab_{sys.argv[1]}.npy"), out)
print(sys.argv[1], float(np.sqrt(np.mean(out.astype(np.float64) ** 2))))
