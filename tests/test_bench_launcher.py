"""`python bench.py --gpus N` must start N ranks by itself: the launcher path on CPU with gloo ranks and
bench.py's stub engine, plus the refusals -- a box with fewer GPUs than asked for, and a rank count that disagrees with
--gpus -- which must fail loudly instead of printing a line for one GPU."""
import json
import os
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, env=env, timeout=timeout)


def test_gpus_2_spawns_two_gloo_ranks_with_the_stub_engine():
    proc = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--streams", "8", "--seconds", "0.2", "--stub-engine"])
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["stub"] is True and line["scaling"] == "weak"
    assert line["steps"] == 2 and line["warmup"] == 1
    # both ranks' shards are in the reduction: 2 ranks x 8 streams x 9600 samples x 2 steps
    assert line["checks"]["total_samples"] == 2 * 8 * 9600 * 2
    assert line["value"] > 0 and line["ms_per_step"] > 0


def test_more_gpus_than_visible_fails_loudly():
    # this container (and a 1-GPU box asked for 2) must not print a line that says n_gpus: 1
    import torch

    if torch.cuda.device_count() >= 2:
        return
    proc = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert proc.returncode != 0
    assert "only" in proc.stderr and "visible" in proc.stderr
    assert not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]


def test_rank_count_must_agree_with_gpus():
    proc = _run(["--gpus", "1", "--stub-engine", "--steps", "1", "--warmup", "0"],
                {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert proc.returncode != 0 and "WORLD_SIZE=2" in proc.stderr


def test_force_distributed_runs_the_collectives_with_one_gloo_rank():
    """--force-distributed: the process group, the barriers and both all-reduces run with a single rank (CPU rehearsal of
    tests/test_gpu_rccl.py, which does the same on the MI355X over RCCL)."""
    proc = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--streams", "8", "--seconds", "0.2", "--stub-engine",
                 "--force-distributed"], {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29731"})
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["collective"]["backend"] == "gloo" and line["collective"]["world_size"] == 1
    assert line["checks"]["total_samples"] == 8 * 9600
