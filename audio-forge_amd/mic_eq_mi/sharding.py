"""Multi-GPU layout of a batch job: contiguous stream ranges per rank, one metric reduction at the end.

Streams are independent -- the reference keeps every piece of DSP state per `AudioProcessor`
 -- so the data path needs no collective: rank g of G owns streams
[g*B/G, (g+1)*B/G), with inputs, outputs and state resident on its own GPU.  The only
exchange is the run's metric vector (SUM over energies / counts, MAX over peaks and gain
reductions): ~100 B, pure latency, two `all_reduce` calls over RCCL (backend "nccl" on ROCm)
or gloo in the CPU tests.
"""
from __future__ import annotations

import numpy as np

SUM_KEYS = ("input_square_sum", "output_square_sum", "true_peak_limited_events", "non_finite_output", "samples")
MAX_KEYS = ("input_sample_peak", "output_sample_peak", "true_peak_limiter_input_peak", "output_true_peak",
            "limiter_peak_gain_reduction_db", "true_peak_limiter_gain_reduction_db",
            "compressor_gain_reduction_db", "deesser_gain_reduction_db", "elapsed_s")


def stream_shard(total_streams: int, rank: int, world: int) -> tuple[int, int]:
    """(first stream, stream count) of `rank`: the contiguous range [rank*B/G, (rank+1)*B/G)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    if total_streams < 0:
        raise ValueError("total_streams must be >= 0")
    first = rank * total_streams // world
    last = (rank + 1) * total_streams // world
    return first, last - first


def local_metrics(rows: np.ndarray, samples: int, elapsed_s: float = 0.0) -> tuple[np.ndarray, np.ndarray]:
    """Fold one rank's block-statistics rows ([blocks, streams] structured array) into the two vectors."""
    sums = np.zeros(len(SUM_KEYS), dtype=np.float64)
    maxes = np.zeros(len(MAX_KEYS), dtype=np.float64)
    for i, key in enumerate(SUM_KEYS):
        sums[i] = float(samples) if key == "samples" else (float(rows[key].astype(np.float64).sum()) if rows.size else 0.0)
    for i, key in enumerate(MAX_KEYS):
        maxes[i] = float(elapsed_s) if key == "elapsed_s" else (float(rows[key].max()) if rows.size else 0.0)
    return sums, maxes


def reduce_metrics(sums: np.ndarray, maxes: np.ndarray, device=None) -> dict:
    """The one collective of a run.  With no process group initialised it is the identity; with one it runs even for a
    single rank (bench.py --force-distributed executes the RCCL path on a one-GPU box)."""
    import torch
    import torch.distributed as dist

    s = torch.as_tensor(np.asarray(sums, dtype=np.float64), device=device).clone()
    m = torch.as_tensor(np.asarray(maxes, dtype=np.float64), device=device).clone()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
    out = {k: float(v) for k, v in zip(SUM_KEYS, s.cpu().tolist())}
    out.update({k: float(v) for k, v in zip(MAX_KEYS, m.cpu().tolist())})
    return out
