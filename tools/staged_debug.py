import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "audio-forge_amd"))
import signals as S
from mic_eq_mi import _lib, mic_eq_core as core
import test_gpu_stages as T

settings = dict(S.limiter_settings(2.0))
audio = S.batch_signal(70, 210) * np.float32(1.6)
n = audio.shape[1]
calls = ((0, 31_007), (31_007, 31_007 + 480 * 77), (31_007 + 480 * 77, n))
want = T.run(core, _lib.KERNEL_PHASED, audio, settings, calls)
got = T.run(core, _lib.KERNEL_STAGED, audio, settings, calls)
print("audio equal:", np.array_equal(want[0].view(np.uint32), got[0].view(np.uint32)))
for name in want[1].dtype.names:
    a, b = want[1][name], got[1][name]
    bad = np.argwhere(a != b)
    if bad.size:
        print(name, "differs in", len(bad), "rows; first:", bad[:6].tolist(), [ (a[tuple(i)], b[tuple(i)]) for i in bad[:6]])
