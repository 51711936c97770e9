"""The execution forms of round 3 against the forms they replaced, bit for bit (audio and block rows): one chain launch per
call following a ready counter (behind the suppressor; behind the systolic EQ when the suppressor is off), the lane-per-stream
EQ kernel, against one chain launch per window with the systolic EQ, and against the EQ inside the chain kernel.  The switches
are read once per process, so every variant is a short child process of `tools/ab_fullchain.py` (70 streams, 2.3 s, two calls,
a coefficient crossfade opening the stream); the parity of the default forms with the oracle is what every other GPU test checks."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "ab_fullchain.py")


def _run(tag: str, mode: str, **env: str) -> None:
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    child_env = dict(os.environ, AB_MODE=mode, **env)
    done = subprocess.run([sys.executable, TOOL, tag], env=child_env, capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-2000:]


def _same(a: str, b: str) -> None:
    done = subprocess.run([sys.executable, TOOL, "cmp", a, b], capture_output=True, text=True, timeout=120)
    assert done.returncode == 0, done.stdout + done.stderr[-1000:]


@pytest.mark.parametrize("mode", ["full", "full+automakeup", "dynamics", "dynamics+automakeup", "full+steep"])
def test_one_launch_forms_equal_the_per_window_forms(mode):
    tag = mode.replace("+", "_")
    tags = [f"t_{tag}_default", f"t_{tag}_per_window", f"t_{tag}_eq_in_chain"]
    try:
        _run(tags[0], mode)
        _run(tags[1], mode, AF_CHAIN_PERSISTENT="0", AF_EQ_STREAM="0")
        _same(tags[0], tags[1])
        if mode.startswith("full"):
            _run(tags[2], mode, AF_EQ_OFFLOAD="0")
            _same(tags[0], tags[2])
    finally:  # (26 MB each: gpurun only brings 64 MiB of gpurun_out/ back)
        for t in tags:
            path = os.path.join(ROOT, "gpurun_out", f"abfc_{t}.npz")
            if os.path.exists(path):
                os.remove(path)
