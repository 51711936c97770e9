"""Latency of short calls (streaming use): 256 streams, N calls of `block` samples each, kernel 2 vs kernel 4."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "audio-forge_amd"))
import signals as S
from mic_eq_mi import _lib, mic_eq_core as core
streams = 256
for block in (480, 960, 4800):
    calls = 200
    x = torch.from_numpy(S.batch_signal(64, (block * calls + 479) // 480)).cuda().repeat(4, 1)[:, : block * calls].contiguous()
    y = torch.empty_like(x)
    for kernel, name in ((_lib.KERNEL_PHASED, "ring"), (_lib.KERNEL_STAGED, "staged")):
        eng = core.Engine(48_000.0, streams)
        core.configure_auto_eq_chain(eng, 48_000.0, S.LIMITER_BANDS, dict(S.limiter_settings(2.0)))
        eng.set_kernel(kernel)
        hs = torch.cuda.current_stream().cuda_stream
        stride = x.shape[1]
        def run():
            for c in range(calls):
                eng.process_device(x.data_ptr() + 4 * c * block, y.data_ptr() + 4 * c * block, block, stride, _lib.LAYOUT_STREAM_MAJOR, hs)
                torch.cuda.synchronize()  # a streaming caller needs each block back
        run()
        eng.reset(); core.configure_auto_eq_chain(eng, 48_000.0, S.LIMITER_BANDS, dict(S.limiter_settings(2.0))); eng.set_kernel(kernel)
        t0 = time.perf_counter(); run(); dt = time.perf_counter() - t0
        print(f"block {block:5d}: {name:6s} {dt / calls * 1e3:7.3f} ms per call", flush=True)
        eng.close()
