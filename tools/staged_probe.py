"""Development probe of the stage pipeline: timing of the dynamics chain, kernel 2 vs kernel 4, at two batch sizes."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "audio-forge_amd"))
import signals as S
from mic_eq_mi import _lib, mic_eq_core as core

TYPED = [("bell", 80.0 * 1.75**i, 3.0 if i % 2 else -2.5, 1.0, 12, True) for i in range(10)]
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
n = int(48_000 * seconds)
for streams in ([int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (256, 4096)):
    base = torch.from_numpy(S.batch_signal(64, int(seconds * 100))).cuda()
    x = base.repeat((streams + 63) // 64, 1)[:streams].contiguous()
    y = torch.empty_like(x)
    for kernel, name in (((_lib.KERNEL_PHASED, "ring"),) if os.environ.get("PROBE_RING", "1") == "1" else ()) + ((_lib.KERNEL_STAGED, "staged"),):
        eng = core.Engine(48_000.0, streams)
        settings = dict(S.limiter_settings(2.0))
        settings["eq_bands_v2"] = TYPED
        if os.environ.get("PROBE_DEESSER"):
            settings["deesser_enabled"] = True
        if os.environ.get("PROBE_ADAPTIVE"):
            settings["compressor_adaptive_release"] = True
        core.configure_auto_eq_chain(eng, 48_000.0, S.LIMITER_BANDS, settings)
        eng.set_kernel(kernel)
        hs = torch.cuda.current_stream().cuda_stream
        times = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.process_device(x.data_ptr(), y.data_ptr(), n, n, _lib.LAYOUT_STREAM_MAJOR, hs)
            t_enq = (time.perf_counter() - t0) * 1e3
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
            times.append(-t_enq)
        print(f"{streams:5d} streams x {seconds:g} s, {name:6s}: " + " ".join((f"{t:8.2f}" if t >= 0 else f"(enqueue {-t:.2f})") for t in times) + " ms", flush=True)
        eng.close()
