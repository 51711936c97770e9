"""RCCL executes on the MI355X (one collective, the metric reduction).  A one-GPU box cannot form a ring of
eight, but everything the N-rank run does around the data path runs here with the rank count there is:
`bench.py --force-distributed` under RANK=0 WORLD_SIZE=1 initialises the `nccl` (= RCCL) process group bound to the device,
runs both barriers and the SUM and MAX all-reduces on device tensors, and prints the line.  The bench is a child process
started before anything in this test touches the GPU for it (never a re-exec of an initialised process)."""
import json
import os
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]

pytestmark = pytest.mark.gpu


def test_rccl_path_runs_with_one_rank_on_the_device():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--streams", "256",
                           "--seconds", "1", "--no-cpu-baseline", "--force-distributed"], capture_output=True, text=True, env=env,
                          timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    line = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0])
    assert line["collective"]["backend"] == "nccl" and line["collective"]["world_size"] == 1
    assert line["n_gpus"] == 1 and line["value"] > 0
    # the reduced vector is the rank's own: 256 streams x 48 000 samples x 2 steps, and a real chain ran
    assert line["checks"]["output_rms"] > 0.01 and line["checks"]["max_compressor_gr_db"] > 0.5
