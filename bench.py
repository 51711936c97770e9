#!/usr/bin/env python3
"""bench.py -- throughput of the batched voice chain on MI355X (one process per GPU).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM: `--streams` independent 48 kHz mono streams x `--seconds` of audio
through the chain (default: BASELINE.json configs[2] shape, batch 4096 x 10 s).  Rank 0
prints ONE JSON line with BASELINE.json's metric (48 kHz mono frames/s, whole job), the
roofline object for the dominant kernel (HIP-event timing taken inside this script on the
stream the kernel runs on) and the CPU baseline (the KAT-pinned oracle timed on this
host's cores on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
for p in (ROOT, ROOT / "audio-forge_amd", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
ALGORITHMIC_BYTES_PER_SAMPLE = 8  # 4 B f32 read + 4 B f32 write
SAMPLE_RATE = 48_000

def _chain_configuration():
    """BASELINE config 2 parameters = the settings dict and bands `evaluation/limiter-lookahead-report.json` was rendered
    with (2 ms lookahead): data captured from the reference's evaluator by tools/gen_golden.py (tests/golden/)."""
    fixture = json.loads((ROOT / "tests" / "golden" / "limiter_lookahead.json").read_text())
    settings = {k: v for k, v in fixture["settings"]["2"].items() if k not in ("return_output_audio", "deesser_enabled")}
    return settings, [tuple(band) for band in fixture["bands"]]


CHAIN_SETTINGS, BANDS = _chain_configuration()


def synth_batch(n_streams: int, n_blocks: int, first_stream: int, device: torch.device) -> torch.Tensor:
    """SURVEY.md 8(d) S3 on the GPU: per-stream reseeded golden-KAT generator, [n_streams, n] f32."""
    from signals import KAT_NOISE_STATE, MASK64, stream_params

    n = n_blocks * 480
    a, c = 6364136223846793005, 1442695040888963407
    mul = np.empty(n, dtype=np.uint64)
    add = np.empty(n, dtype=np.uint64)
    m, d = 1, 0
    for k in range(n):  # affine powers of the LCG: state_k = mul[k]*s0 + add[k]
        m = (m * a) & MASK64
        d = (d * a + c) & MASK64
        mul[k] = m
        add[k] = d
    mul_t = torch.from_numpy(mul.view(np.int64)).to(device)
    add_t = torch.from_numpy(add.view(np.int64)).to(device)
    idx = torch.arange(n, device=device, dtype=torch.float64)
    t = idx / 48000.0
    gate = (((torch.arange(n, device=device) // 480) // 12) % 5 == 2).to(torch.float64)
    sib = gate * 0.35 * torch.sin(2.0 * np.pi * 7200.0 * t)
    out = torch.empty((n_streams, n), dtype=torch.float32, device=device)
    two_pi_t = 2.0 * np.pi * t
    for s in range(n_streams):
        state, f0, ph = stream_params(first_stream + s)
        s0 = torch.tensor(np.array([state & MASK64], dtype=np.uint64).view(np.int64), device=device)
        states = mul_t * s0 + add_t  # wrapping int64 arithmetic == uint64 LCG
        hi = ((states >> 40) & 0xFFFFFF).to(torch.float64)
        noise = (hi / float((1 << 24) - 1) * 2.0 - 1.0) * 0.012
        phrase = 0.25 + 0.75 * torch.abs(torch.sin(two_pi_t * ph))
        voiced = 0.30 * torch.sin(two_pi_t * f0) + 0.14 * torch.sin(two_pi_t * (2.0 * f0)) + 0.08 * torch.sin(two_pi_t * (15.0 * f0))
        out[s] = (phrase * voiced + sib + noise).to(torch.float32)
    return out


CLOCK_HZ = 2.4e9  # MI355X_MICROARCH.md: max shader clock (the issue roof below is priced at it)


def profile_counters(full: bool, streams: int, seconds: float, auto_makeup: bool = False, deesser: bool = False):
    """Counter figures of this workload from the committed rocprofv3 --pmc passes (profiles/r*_*counters*.json, written by
    tools/step_counters.py from separate counter runs of this same command; counters cannot be read from inside a timed
    run).  The newest round's file taken at exactly this shape is used; it carries the commit it was taken at."""
    best = None
    for path in sorted((ROOT / "profiles").glob("r[0-9][0-9]_*counters*.json")):
        try:
            prof = json.loads(path.read_text())
        except (OSError, ValueError):
            continue
        shape = prof.get("shape") or {}
        if (shape.get("streams") == streams and shape.get("seconds") == seconds and shape.get("chain") == ("full" if full else "dynamics")
                and bool(shape.get("auto_makeup", False)) == auto_makeup and bool(shape.get("deesser", False)) == deesser
                and "kernels" in prof):
            prof["_file"] = f"profiles/{path.name}"
            best = prof  # (sorted by name: the highest round wins)
    return best


F32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate), = the f32 vector peak
RNN_LAYERS_MNK = [  # the network's contractions per frame, M = streams (~86.9 k MAC = 174 kFLOP per frame per stream)
    ("dense 42->24", 24, 42), ("GRU 24 <- 24+24", 72, 48), ("GRU 48 <- 90+48", 144, 138), ("GRU 96 <- 114+96", 288, 210),
    ("dense 96->22", 22, 96),
]


def mfma_block(prof: dict, streams: int, frames: int) -> dict | None:
    """roofline.mfma ("MFMA utilisation reported for that kernel alone ... with M/N/K shapes"): the RNNoise
    network kernel's matrix-core figures from the counter passes of this workload."""
    row = next((v for k, v in prof["kernels"].items() if "supp_rnn_kernel" in k), None)
    if row is None:
        return None
    sq = row.get("sq", {})
    macs = sum(n * k for _name, n, k in RNN_LAYERS_MNK)
    flops_step = 2.0 * macs * streams * frames
    out = {
        "kernel": "supp_rnn_kernel<4>", "instruction": "v_mfma_f32_16x16x4_f32 (exact f32 fma chains; bf16 operands would miss the 1e-5 budget)",
        "shapes_MNK": [{"layer": name, "M": streams, "N": n, "K": k} for name, n, k in RNN_LAYERS_MNK],
        "algorithmic_flops_per_step": flops_step, "launches_per_step": row["launches_per_step"],
        "peak": F32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
    }
    wave_cycles = sq.get("SQ_WAVE_CYCLES")
    busy = sq.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if busy is not None:
        out["mfma_busy_cycles_per_step"] = busy
        if wave_cycles:
            out["mfma_busy_over_wave_lifetime"] = busy / (4.0 * wave_cycles)  # SQ_WAVE_CYCLES counts quad-cycles (guide)
    if sq.get("SQ_INSTS_VALU_MFMA_MOPS_F32") is not None:
        out["mfma_mops_f32_per_step"] = sq["SQ_INSTS_VALU_MFMA_MOPS_F32"]
    if row.get("kernel_ms_per_launch"):
        t = row["kernel_ms_per_launch"] * 1e-3 * row["launches_per_step"]
        out["achieved"] = flops_step / t / 1e12
        out["frac"] = out["achieved"] / F32_MATRIX_PEAK_TFLOPS
        if busy is not None:
            out["mfma_busy_frac_of_chip"] = busy / (t * CLOCK_HZ * 1024)  # 256 CUs x 4 SIMDs
    return out


def _percentile(values, q: float) -> float:
    v = np.sort(np.asarray(values, dtype=np.float64))
    pos = (v.size - 1) * q
    lo, hi = int(np.floor(pos)), int(np.ceil(pos))
    return float(v[lo] + (pos - lo) * (v[hi] - v[lo]))


def cpu_baseline(seconds: float, full: bool, chain_settings: dict | None = None) -> dict:
    """The CPU restatement of rust-core (oracle/, KAT-pinned; the Rust reference cannot be built offline) timed on this
    host, per SURVEY.md 8(d):
      single_thread: one S1 stream x `seconds`, 1 warm-up + 7 repetitions, median and p95
                     (python/tools/evaluate_limiter_lookahead.py:28,288-289,319-323);
      gpu_share_threads: S3 batch of 256 streams x 2 s, one stream shard per thread on the host cores that belong to ONE GPU's
                     share of the box (min(sched_getaffinity, 16): a one-GPU box is given 16 of the host's cores; ctypes releases
                     the GIL), 1 warm-up + 7 repetitions, median and p95 -- NOT the whole host (`host_cores` says how many it has);
      suppressor:    per-frame time percentiles of the RNNoise stage alone (bin/rnnoise_benchmark.rs:92-96).
    The RNNoise stage runs its packed mixed-radix transforms (afo_rnn_fft_mode = 1, same results to the last bits)."""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor

    sys.path.insert(0, str(ROOT / "oracle"))
    import af_oracle_py as oracle  # the checker / baseline, never the product path
    from signals import kat_signal, stream_params

    settings = dict(chain_settings or CHAIN_SETTINGS)
    lib = oracle.lib()
    ctypes.c_int.in_dll(lib, "afo_rnn_fft_mode").value = 1

    def make_input(index: int, n_blocks: int) -> np.ndarray:
        st, f0, ph = stream_params(index)
        return kat_signal(n_blocks, st, f0, ph)

    def run_stream(x: np.ndarray) -> int:  # the timed work: only calls into the C restatement
        if full:
            x = oracle.suppressor_process(oracle.prefilter(x), 1.0, 0x5EED)
        oracle.simulate_auto_eq_chain(x, SAMPLE_RATE, BANDS, settings)
        return x.size

    try:
        # ---- one thread, one stream
        n_blocks = int(seconds * 100)
        run_stream(make_input(0, 20))  # tables, page-in
        x0 = make_input(0, n_blocks)
        run_stream(x0)  # warm-up
        times = []
        for _ in range(7):
            t0 = time.perf_counter()
            frames = run_stream(x0)
            times.append(time.perf_counter() - t0)
        single = {"value": frames / _percentile(times, 0.5), "unit": "frames/s", "cores": 1, "repetitions": 7,
                  "median_s": _percentile(times, 0.5), "p95_s": _percentile(times, 0.95),
                  "x_realtime": frames / _percentile(times, 0.5) / SAMPLE_RATE,
                  "sample": f"1 stream x {seconds:g} s (S1), 1 warm-up + 7 repetitions"}
        # ---- the host cores of this GPU's share: thread per stream shard (inputs are generated before the clock starts)
        try:
            available = len(os.sched_getaffinity(0))
        except AttributeError:
            available = os.cpu_count() or 1
        threads = max(1, min(available, 16))  # a one-GPU box's CPU share is 16 cores
        batch, shard_blocks = 256, 200
        inputs = [make_input(i, shard_blocks) for i in range(batch)]
        shards = [inputs[t::threads] for t in range(threads)]

        def run_shard(xs):
            return sum(run_stream(x) for x in xs)

        times = []
        with ThreadPoolExecutor(max_workers=threads) as pool:
            for rep in range(8):
                t0 = time.perf_counter()
                total = sum(pool.map(run_shard, shards))
                if rep:
                    times.append(time.perf_counter() - t0)
        multi = {"value": total / _percentile(times, 0.5), "unit": "frames/s", "cores": threads, "repetitions": 7,
                 "median_s": _percentile(times, 0.5), "p95_s": _percentile(times, 0.95),
                 "x_realtime": total / _percentile(times, 0.5) / SAMPLE_RATE, "cores_available": available,
                 "sample": f"{batch} streams x {shard_blocks / 100:g} s (S3), one shard per thread on {threads} threads "
                           f"(one GPU's share of the host), 1 warm-up + 7 repetitions"}
        # ---- RNNoise stage per frame
        frame_stats = None
        if full:
            st = ctypes.create_string_buffer(1 << 18)
            lib.afo_suppressor_init(st, 1.0, ctypes.c_uint64(0x5EED))
            lib.afo_suppressor_process_frame.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
            x = kat_signal(600)
            out = np.zeros(480, dtype=np.float32)
            fp = ctypes.POINTER(ctypes.c_float)
            per_frame = []
            for f in range(600):
                frame = np.ascontiguousarray(x[f * 480 : (f + 1) * 480])
                t0 = time.perf_counter_ns()
                lib.afo_suppressor_process_frame(st, out.ctypes.data_as(fp), frame.ctypes.data_as(fp))
                per_frame.append((time.perf_counter_ns() - t0) * 1e-9)
            per_frame = per_frame[100:]
            frame_stats = {"frames": len(per_frame), "mean_s": float(np.mean(per_frame)), "p95_s": _percentile(per_frame, 0.95),
                           "p99_s": _percentile(per_frame, 0.99), "max_s": float(np.max(per_frame)),
                           "reference_published": "nnnoiseless 0.5.2 on Zen 4: 40.9 us/frame, p95 50.9, p99 70.0 (BASELINE.md)"}
    finally:
        ctypes.c_int.in_dll(lib, "afo_rnn_fft_mode").value = 0
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": multi["value"], "unit": "frames/s", "cores": multi["cores"], "kind": "port",
        "sample": multi["sample"] + f"; same chain ({'full' if full else 'dynamics'}); CPU restatement of rust-core (oracle/), "
                  "the Rust reference cannot be built offline",
        "x_realtime": multi["x_realtime"], "single_thread": single, "gpu_share_threads": multi, "suppressor_per_frame": frame_stats,
        "host_cpu": cpu_model, "host_cores": os.cpu_count(),
    }


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU, LOCAL_RANK = RANK) with a
    fresh rendezvous port and wait for them.  Runs BEFORE anything in this process touches a GPU (the parent never does:
    counting devices does not initialise HIP), so no initialised process is ever re-executed.  Rank 0 prints the JSON line;
    the return code is non-zero when any rank failed."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(rank), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this platform (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, str(pathlib.Path(__file__).resolve()), *sys.argv[1:]], env=env))
    worst = 0
    for rank, proc in enumerate(procs):
        rc = proc.wait()
        if rc != 0:
            print(f"bench.py: rank {rank} exited with code {rc}", file=sys.stderr)
            worst = worst or (rc if rc > 0 else 1)
    return worst


class StubEngine:
    """CPU rehearsal of the launch / sharding / reduction path (`--stub-engine`, tests/test_bench_launcher.py): no kernels,
    block rows carry the input's own energy and peak.  A line produced with it is marked `"stub": true` and is not a
    measurement."""

    def __init__(self, streams: int, n: int, first_stream: int):
        from mic_eq_mi import mic_eq_core as core

        rng = np.random.default_rng(1234 + first_stream)
        self.x = (rng.standard_normal((streams, n)) * 0.1).astype(np.float32)
        blocks = n // 960
        self.rows = np.zeros((blocks, streams), dtype=core.STATS_DTYPE)
        xb = self.x[:, : blocks * 960].reshape(streams, blocks, 960).astype(np.float64)
        self.rows["input_square_sum"] = (xb ** 2).sum(axis=2).T
        self.rows["output_square_sum"] = self.rows["input_square_sum"]
        self.rows["input_sample_peak"] = np.abs(xb).max(axis=2).T
        self.rows["output_sample_peak"] = self.rows["input_sample_peak"]

    def step(self) -> None:
        float(np.square(self.x, dtype=np.float32).sum())


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=4096, help="streams PER GPU (weak scaling)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--variant", type=str, default="", help="lane | ring-<waves>x<chunk> (tuning)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chain", choices=["full", "dynamics"], default="full",
                    help="full = DC/HP prefilter + RNNoise + EQ + compressor + limiter + true-peak (configs[2]); "
                         "dynamics = EQ + compressor + limiter + true-peak only (configs[1] chain)")
    ap.add_argument("--auto-makeup", action="store_true",
                    help="compressor auto-makeup on (north_star's chain as literally named: per-block activity -> momentary "
                         "loudness -> makeup, compressor.rs:598-653); target -16 LUFS")
    ap.add_argument("--deesser", action="store_true",
                    help="three-band de-esser ahead of the EQ (deesser.rs:405-547; the golden KAT's settings: auto amount 0.85, "
                         "max reduction 10 dB) -- SURVEY 8(f) row 1, not part of BASELINE's configs")
    ap.add_argument("--force-distributed", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the barrier and both metric all-reduces on device tensors "
                         "even with one rank (RANK=0 WORLD_SIZE=1): executes the collective path on a one-GPU box")
    ap.add_argument("--stub-engine", action="store_true",
                    help="CPU rehearsal of the N-rank launch path: gloo ranks and a stub engine; not a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become the launcher (nothing has touched a GPU yet)
        if not args.stub_engine:
            visible = torch.cuda.device_count()  # does not initialise HIP on this image
            if visible < args.gpus:
                raise SystemExit(f"bench.py: --gpus {args.gpus} but only {visible} GPU(s) are visible; refusing to print a "
                                 f"line for fewer ranks than asked for")
        raise SystemExit(launch_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    distributed = world > 1 or args.force_distributed
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself; under torchrun pass --nproc-per-node N)")
    if args.stub_engine:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if args.stub_engine:
            dist.init_process_group("gloo", rank=rank, world_size=world)  # (CPU rehearsal of the launch / reduction path)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    if args.variant:
        os.environ["AF_KERNEL_VARIANT"] = args.variant
    import mic_eq_mi
    from mic_eq_mi import mic_eq_core as core
    from mic_eq_mi import sharding

    if not mic_eq_mi.CORE_AVAILABLE:
        raise SystemExit("libaudioforge_mi.so is missing; run __graft_entry__.build()")

    n_blocks = int(round(args.seconds * 100))
    n = n_blocks * 480
    streams = args.streams
    first_stream, shard = sharding.stream_shard(world * streams, rank, world)  # weak scaling: B/G fixed per GPU
    assert shard == streams
    if args.stub_engine:
        return run_stub(args, world, rank, streams, n, first_stream)
    x = synth_batch(streams, n_blocks, first_stream, device)
    y = torch.empty_like(x)
    torch.cuda.synchronize()

    engine = core.Engine(SAMPLE_RATE, streams, local_rank)
    chain_settings = dict(CHAIN_SETTINGS)
    if args.auto_makeup:
        chain_settings.update(compressor_auto_makeup_enabled=True, compressor_target_lufs=-16.0)
    if args.deesser:
        chain_settings.update(deesser_enabled=True, deesser_auto_enabled=True, deesser_auto_amount=0.85, deesser_max_reduction_db=10.0)
    core.configure_auto_eq_chain(engine, float(SAMPLE_RATE), BANDS, chain_settings)
    if not args.variant:
        engine.set_kernel(args.kernel)
    full = args.chain == "full"
    if full:
        engine.set_prefilter_enabled(1, 1)   # DC block + 80 Hz high-pass, routing.rs:826-843
        engine.set_suppressor_enabled(1)     # RNNoise, rnnoise.rs:122-164 (synthetic weights: trained ones are not offline)
    engine.set_timing_enabled(1)
    # (torch's current stream, i.e. the default one: a stream of the caller's own is one more hardware queue for the process, and the
    # engine's eight are at the point where one more makes two pipeline stages share a queue -- 172 -> 177 ms per step)
    hip_stream = torch.cuda.current_stream().cuda_stream

    def step() -> None:
        engine.process_device(x.data_ptr(), y.data_ptr(), n, n, core._lib.LAYOUT_STREAM_MAJOR, hip_stream)

    def barrier() -> None:
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms, supp_ms, chain_ms, first_ms, tail_ms, segments = [], [], [], [], [], 1
    sm, cm = C.c_double(0.0), C.c_double(0.0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(engine.last_kernel_ms()[0])  # waits on this step's stop event only
        engine._lib.af_engine_last_stage_ms(engine._h, C.byref(sm), C.byref(cm))
        supp_ms.append(sm.value)
        chain_ms.append(cm.value)
        f_ms, t_ms, segments = engine.last_chain_launch_ms()
        first_ms.append(f_ms)
        tail_ms.append(t_ms)
    barrier()
    elapsed = time.perf_counter() - t0

    # the one collective: metric reduction over ranks (RCCL over xGMI when world > 1)
    rows = engine.block_stats()
    sums, maxes = sharding.local_metrics(rows, streams * n * args.steps, elapsed)
    merged = sharding.reduce_metrics(sums, maxes, device)
    elapsed_max = merged["elapsed_s"]
    total_frames = int(merged["samples"])
    value = total_frames / elapsed_max
    collective = None
    if distributed:
        collective = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                      "calls": "barrier x2, all_reduce(SUM) + all_reduce(MAX) over a 14-double vector on the device"}

    used = int(engine._lib.af_engine_last_kernel(engine._h))
    ring = args.variant[5:] if args.variant.startswith("ring-") else "16x4"
    quad = args.variant[5:] if args.variant.startswith("quad-") else "12"
    kernel_name = {1: "chain_lane_kernel", 2: f"chain_ring_kernel<{ring}>", 3: f"chain_quad_kernel<{quad}>",
                   4: "stage_diag_serial_kernel (the serial stages of one launch step of the stage pipeline)",
                   5: "chain_comp_roles_kernel + chain_lim_roles_kernel (role pipeline)"}.get(used, "?")
    if rank == 0:
        # dominant kernel: the chain launch (HIP events recorded by the engine around it on the stream it runs on)
        # (behind the suppressor the chain is ONE launch per call that follows the windows as they arrive -- its duration then
        # includes what it waits for them; configurations that keep one launch per window report their average)
        launches = max(1, int(segments))
        avg_kernel_s = float(np.mean(first_ms)) / 1000.0 / launches
        frames_per_launch = streams * n // launches
        achieved = ALGORITHMIC_BYTES_PER_SAMPLE * frames_per_launch / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        chain_groups = (streams + 63) // 64 if used == 2 else ((streams + 15) // 16 if used == 3 else (streams + 63) // 64)
        roofline = {
            # scan/IIR/FIR work is priced against HBM by convention; what actually bounds this kernel is
            # vector issue on the CUs its workgroups occupy (see `compute` below and DESIGN.md 4.3)
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "kernel": kernel_name, "avg_kernel_ms": avg_kernel_s * 1000.0, "launches_per_step": launches,
            "algorithmic_bytes_per_launch": ALGORITHMIC_BYTES_PER_SAMPLE * frames_per_launch,
            "limiting_resource": ("vector issue on the chain's CUs (serial recurrences: one 16-wave workgroup per 64 streams per CU)" if used != 4 else
                                  "dependent-instruction latency of one wave per recurrence (~8 cycles per vector instruction, "
                                  "tools/probe/valu_latency.hip): a step lasts as long as its longest stage, the EQ"),
        }
        prof = profile_counters(full, streams, args.seconds, args.auto_makeup, args.deesser)
        if prof is not None:
            row = next((v for k, v in prof["kernels"].items() if kernel_name.split("<")[0].split(" ")[0] in k), None)
            if row is not None:
                # the file holds per-STEP totals; a launch of THIS run is 1 / launches of a step (the counter passes themselves
                # run one chain launch per window: rocprofv3 --pmc serialises dispatches, and a launch that follows its
                # producers' counter needs them beside it -- the engine falls back by itself there)
                per_launch = launches
                roofline["traffic"] = (row["fetch_bytes"] + row["write_bytes"]) / per_launch  # HBM bytes per launch (PMC)
                cus = min(chain_groups, 256)
                t_kernel = avg_kernel_s
                roofline["compute"] = {
                    "cus_used": cus, "waves_per_simd": 4 if used == 2 else None,
                    "valu_insts_per_launch": row["valu_insts"] / per_launch,
                    # wave-instructions issued / (CUs x 4 SIMDs x clock x time): one instruction per SIMD-cycle as the roof
                    # (a SIMD-32 issues a wave64 f32 instruction over 2 cycles, an f64 one over 4)
                    "valu_issue_frac": row["valu_insts"] / per_launch / (cus * 4 * CLOCK_HZ * t_kernel),
                    "valu_issue_frac_whole_chip": row["valu_insts"] / per_launch / (256 * 4 * CLOCK_HZ * t_kernel),
                    "achieved_f64_tflops": row.get("f64_flops", 0.0) / per_launch / t_kernel / 1e12,
                    "peak_f64_tflops": 78.6,
                }
            roofline["step_traffic"] = {
                "fetch_bytes": sum(v["fetch_bytes"] for v in prof["kernels"].values()),
                "write_bytes": sum(v["write_bytes"] for v in prof["kernels"].values()),
                "algorithmic_bytes": ALGORITHMIC_BYTES_PER_SAMPLE * streams * n,
                "per_kernel": {k: v["fetch_bytes"] + v["write_bytes"] for k, v in prof["kernels"].items()},
            }
            if full:
                roofline["mfma"] = mfma_block(prof, streams, n // 480)
            roofline["counters_from"] = {"file": prof["_file"], "commit": prof.get("commit"),
                                         "note": "separate rocprofv3 --pmc runs of this command (per-step totals; taken with one chain launch per "
                                                 "window, which is what the engine does under a serialising profiler); not measured in this run"}
        line = {
            "metric": "48 kHz mono frames/s (real-time-factor x streams), voice chain",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1000.0,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "x_realtime": value / SAMPLE_RATE,
            "config": {
                "workload": (f"batch={streams} streams/GPU x {args.seconds:g} s @48 kHz, full chain: DC block + 80 Hz HP -> RNNoise "
                             f"suppressor (synthetic weights) -> 10-band EQ -> compressor{' with auto-makeup' if args.auto_makeup else ''} "
                             f"-> 2 ms lookahead limiter -> 4x true-peak limiter/detector (BASELINE configs[2])") if full else
                            (f"batch={streams} streams/GPU x {args.seconds:g} s @48 kHz, 10-band EQ + compressor + 2 ms lookahead "
                             f"limiter + 4x true-peak limiter/detector, no suppressor (BASELINE configs[1] chain)"),
                "streams_per_gpu": streams, "seconds": args.seconds, "control_block": 960, "layout": "stream-major",
                "kernel": kernel_name, "sharding": f"streams x{world}, no data-path collective",
                "auto_makeup": bool(args.auto_makeup), "deesser": bool(args.deesser),
            },
            "collective": collective,
            "roofline": roofline,
            "stage_ms": {"suppressor_and_front_end": float(np.mean(supp_ms)), "chain": float(np.mean(chain_ms)),
                         "all_kernels": float(np.mean(kernel_ms))},
            "checks": {"output_rms": float(np.sqrt(merged["output_square_sum"] / (world * streams * n))),
                       "output_sample_peak": merged["output_sample_peak"], "output_true_peak": merged["output_true_peak"],
                       "max_compressor_gr_db": merged["compressor_gain_reduction_db"],
                       "true_peak_limited_blocks": int(merged["true_peak_limited_events"])},
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a single-GPU-run artefact (rank 0, N = 1)
            line["cpu_baseline"] = cpu_baseline(args.seconds, full, chain_settings)
        elif world > 1:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    engine.close()
    if distributed:
        dist.destroy_process_group()


def run_stub(args, world: int, rank: int, streams: int, n: int, first_stream: int) -> None:
    """The launch path with a stub engine on CPU (gloo): same barrier / timing / reduction code, no kernels."""
    import torch.distributed as dist

    from mic_eq_mi import sharding

    engine = StubEngine(streams, n, first_stream)

    def barrier() -> None:
        if world > 1 or args.force_distributed:
            dist.barrier()

    for _ in range(args.warmup):
        engine.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        engine.step()
    barrier()
    elapsed = time.perf_counter() - t0
    sums, maxes = sharding.local_metrics(engine.rows, streams * n * args.steps, elapsed)
    merged = sharding.reduce_metrics(sums, maxes, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({
            "metric": "48 kHz mono frames/s (real-time-factor x streams), voice chain", "stub": True,
            "value": merged["samples"] / merged["elapsed_s"], "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": merged["elapsed_s"] / args.steps * 1000.0, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "STUB ENGINE (launch-path rehearsal on CPU, not a measurement)", "streams_per_gpu": streams,
                       "sharding": f"streams x{world}, no data-path collective"},
            "collective": ({"backend": dist.get_backend(), "world_size": dist.get_world_size()} if dist.is_initialized() else None),
            "checks": {"total_samples": int(merged["samples"]), "input_square_sum": merged["input_square_sum"]},
        }))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
