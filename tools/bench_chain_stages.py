"""Per-stage cost of the chain kernel: time the ring kernel with stage subsets (batch 4096 x 2 s)."""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "audio-forge_amd"), str(ROOT / "tests")]
import torch
import bench
from mic_eq_mi import mic_eq_core as core

B, blocks = 4096, 200
dev = torch.device("cuda", 0)
x = bench.synth_batch(B, blocks, 0, dev)
y = torch.empty_like(x)
n = x.shape[1]
for name, eq, comp, lim in (("all", 1, 1, 1), ("eq", 1, 0, 0), ("comp", 0, 1, 0), ("limiter+tp", 0, 0, 1), ("none", 0, 0, 0),
                            ("eq+comp", 1, 1, 0), ("comp+limiter+tp", 0, 1, 1)):
    eng = core.Engine(48000.0, B, 0)
    core.configure_auto_eq_chain(eng, 48000.0, bench.BANDS, bench.CHAIN_SETTINGS)
    eng.set_eq_enabled(eq); eng.set_compressor_enabled(comp); eng.set_limiter_enabled(lim)
    eng.set_timing_enabled(1)
    ms = []
    for i in range(3):
        eng.process_device(x.data_ptr(), y.data_ptr(), n, n, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ms.append(eng.last_kernel_ms()[0])
    print(f"{name:12s} {min(ms):8.2f} ms  ({B*n/min(ms)*1e3/48000:.0f}x RT)")
    eng.close()
