"""Does the two-part EQ run on a non-default caller stream at 4096 streams?  (probe for the null-stream deadlock)"""
import sys, pathlib, time
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "audio-forge_amd"), str(ROOT / "tests")]
import torch
import bench
from mic_eq_mi import mic_eq_core as core

B, blocks = 4096, 1000
dev = torch.device("cuda", 0)
x = bench.synth_batch(B, blocks, 0, dev)
y = torch.empty_like(x)
n = x.shape[1]
use_null = len(sys.argv) > 1 and sys.argv[1] == "null"
st = torch.cuda.current_stream() if use_null else torch.cuda.Stream()
eng = core.Engine(48000.0, B, 0)
core.configure_auto_eq_chain(eng, 48000.0, bench.BANDS, bench.CHAIN_SETTINGS)
eng.set_prefilter_enabled(1, 1)
eng.set_suppressor_enabled(1)
torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter()
    eng.process_device(x.data_ptr(), y.data_ptr(), n, n, 0, st.cuda_stream)
    st.synchronize()
    print("null" if use_null else "own stream", i, round((time.perf_counter() - t0) * 1e3, 1), "ms", flush=True)
rows = eng.block_stats()
print("ok", float(y.float().pow(2).mean().sqrt()))
eng.close()
