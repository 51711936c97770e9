"""The N>1 path on CPU: two gloo ranks shard a batch by stream range, process their shards
independently and meet in the metric reduction; the result equals the unsharded run.

The HIP engine cannot run here, so each rank's shard is rendered by the CPU oracle -- the test is about
the sharding arithmetic and the collective (the same code bench.py runs on RCCL), not about the kernels.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import signals as S

TOTAL_STREAMS = 5  # ragged on purpose: shards of 2 and 3
N_BLOCKS = 25


def _render(first, count):
    import af_oracle_py as oracle

    rows = np.zeros((N_BLOCKS // 2 + 1, count), dtype=[(k, "<f8") for k in (
        "input_square_sum", "output_square_sum", "true_peak_limited_events", "non_finite_output", "input_sample_peak",
        "output_sample_peak", "true_peak_limiter_input_peak", "output_true_peak", "limiter_peak_gain_reduction_db",
        "true_peak_limiter_gain_reduction_db", "compressor_gain_reduction_db", "deesser_gain_reduction_db")])
    for j in range(count):
        x = S.kat_signal(N_BLOCKS, *S.stream_params(first + j))
        chain = oracle.Chain(48_000.0)
        chain.set("compressor_enabled", 1)
        y = x.copy()
        for b, pos in enumerate(range(0, x.size, 960)):
            blk_in = x[pos : pos + 960]
            st = chain.process_block(y[pos : pos + 960])
            rows["input_square_sum"][b, j] = float(np.sum(blk_in.astype(np.float64) ** 2))
            rows["output_square_sum"][b, j] = float(np.sum(y[pos : pos + 960].astype(np.float64) ** 2))
            rows["input_sample_peak"][b, j] = float(np.abs(blk_in).max())
            rows["output_sample_peak"][b, j] = float(np.abs(y[pos : pos + 960]).max())
            for key in ("true_peak_limited_events", "output_true_peak", "limiter_peak_gain_reduction_db",
                        "true_peak_limiter_gain_reduction_db", "compressor_gain_reduction_db", "deesser_gain_reduction_db"):
                rows[key][b, j] = float(getattr(st, key))
            rows["true_peak_limiter_input_peak"][b, j] = float(st.true_peak_limiter_input_peak)
    return rows, count * N_BLOCKS * 480


def _worker(rank, world, port, queue):
    import torch.distributed as dist

    from mic_eq_mi import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, count = sharding.stream_shard(TOTAL_STREAMS, rank, world)
        rows, samples = _render(first, count)
        sums, maxes = sharding.local_metrics(rows, samples, elapsed_s=1.0 + rank)
        dist.barrier()
        merged = sharding.reduce_metrics(sums, maxes)
        queue.put((rank, first, count, merged))
    finally:
        dist.destroy_process_group()


def test_stream_shard_ranges_tile_the_batch():
    from mic_eq_mi import sharding

    for total in (0, 1, 5, 256, 4096, 32768, 1001):
        for world in (1, 2, 3, 4, 8):
            spans = [sharding.stream_shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert spans[-1][0] + spans[-1][1] == total
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        sharding.stream_shard(8, 2, 2)


def test_two_rank_gloo_run_equals_single_process(oracle):
    from mic_eq_mi import sharding

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(queue.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in results] == [(0, 2), (2, 3)]
    assert results[0][3] == results[1][3]  # every rank holds the same reduced vector

    rows, samples = _render(0, TOTAL_STREAMS)
    sums, maxes = sharding.local_metrics(rows, samples, elapsed_s=2.0)
    want = sharding.reduce_metrics(sums, maxes)  # no process group here: identity
    got = results[0][3]
    for key in sharding.MAX_KEYS + ("true_peak_limited_events", "non_finite_output", "samples"):
        assert got[key] == want[key], key
    for key in ("input_square_sum", "output_square_sum"):  # summation order differs across shards
        assert abs(got[key] - want[key]) <= 1e-12 * want[key], key
    assert got["samples"] == TOTAL_STREAMS * N_BLOCKS * 480
    assert got["elapsed_s"] == 2.0  # MAX over ranks of (1 + rank)
