"""The pins the reference itself holds for the RNNoise wrapper (a3/a4), on the CPU oracle and on the host-only part of
the C ABI.  Each test restates one reference test (cited): rust-core/src/dsp/rnnoise.rs:330-451,
rust-core/src/audio/processor/tests.rs:1887-1928, rust-core/src/dsp/noise_suppressor.rs:196-220.

The core (`nnnoiseless::DenoiseState::process_frame`) stays PARITY-UNPINNED (crate and weights absent); what is pinned
here is everything around it -- the model-input transfer function, the frame rings and their counts, strength clamp,
bypass -- plus one implementation-independent check of the core restatement: its wavefront-native evaluation order
(what the GPU computes) against the published scalar C's running sums.
"""
import ctypes as C

import numpy as np
import pytest

import signals as S

PCM_SCALE = 32768.0
PCM_MODEL_LIMIT = 32760.0
SOFT_CLIP_THRESHOLD = 0.98


def check_soft_clip_transfer(scale):
    """test_rnnoise_model_input_soft_clip_transfer, rnnoise.rs:335-352 (`scale` maps one float to one float)."""
    below = scale(0.5)
    assert abs(below - 0.5 * PCM_SCALE) < 1e-3
    near_full_scale = scale(1.0)
    assert near_full_scale > SOFT_CLIP_THRESHOLD * PCM_SCALE
    assert near_full_scale < PCM_MODEL_LIMIT
    louder = scale(1.5)
    assert louder > near_full_scale
    assert louder <= PCM_MODEL_LIMIT
    negative = scale(-1.0)
    assert abs(negative + near_full_scale) < 1e-3
    assert scale(float("nan")) == 0.0
    # beyond the reference's five points: +-inf -> 0, monotone above the knee, odd symmetry, never past the model limit
    assert scale(float("inf")) == 0.0 and scale(float("-inf")) == 0.0
    xs = np.linspace(0.98, 40.0, 4001, dtype=np.float32)
    ys = np.array([scale(float(x)) for x in xs])
    assert np.all(np.diff(ys) >= 0.0) and ys[-1] <= PCM_MODEL_LIMIT and ys[0] == np.float32(0.98) * np.float32(PCM_SCALE)
    for x in (0.3, 0.981, 1.0, 7.5):
        assert scale(-x) == -scale(x)


def test_soft_clip_transfer_pins(oracle):
    check_soft_clip_transfer(oracle.scale_sample_for_model)


def test_frame_buffering_counts(oracle):
    """test_rnnoise_frame_buffering, rnnoise.rs:355-370."""
    p = oracle.RNNoiseProcessor()
    p.push_samples(np.zeros(400, dtype=np.float32))
    p.process_frames()
    assert p.available_samples() == 0
    assert p.pending_input() == 400
    p.push_samples(np.zeros(100, dtype=np.float32))
    p.process_frames()
    assert p.available_samples() == 480
    assert p.pending_input() == 20


def test_bypass_and_hot_samples(oracle):
    """test_rnnoise_bypass + test_disabled_rnnoise_preserves_hot_input_samples, rnnoise.rs:372-399."""
    p = oracle.RNNoiseProcessor()
    p.set_enabled(False)
    p.push_samples(np.ones(100, dtype=np.float32))
    p.process_frames()
    assert p.available_samples() == 100
    p = oracle.RNNoiseProcessor()
    p.set_enabled(False)
    hot = np.array([1.25, -1.5, 0.25, -0.75], dtype=np.float32)
    p.push_samples(hot)
    p.process_frames()
    out = p.read_samples(4)
    assert out.size == 4 and np.array_equal(out, hot)


def test_strength_getter_setter_and_mix(oracle):
    """test_rnnoise_strength_getter_setter + test_rnnoise_wet_dry_mix, rnnoise.rs:401-434."""
    p = oracle.RNNoiseProcessor()
    assert p.get_strength() == 1.0
    p.set_strength(0.5)
    assert p.get_strength() == 0.5
    p.set_strength(1.5)
    assert p.get_strength() == 1.0
    p.set_strength(-0.5)
    assert p.get_strength() == 0.0
    p = oracle.RNNoiseProcessor(0.5)
    p.push_samples(np.full(480, 0.5, dtype=np.float32))
    p.process_frames()
    assert p.available_samples() == 480


def test_clipped_input_stays_finite(oracle):
    """test_rnnoise_output_stays_finite_for_clipped_input, rnnoise.rs:436-450."""
    p = oracle.RNNoiseProcessor()
    for n in range(480):
        p.push_samples(np.array([1.0 if n % 2 == 0 else -1.0], dtype=np.float32))
    p.process_frames()
    out = p.read_samples(480)
    assert out.size == 480
    assert np.all(np.isfinite(out)) and float(np.abs(out).max()) <= 2.0


def test_full_rt_block_without_short_write(oracle):
    """test_rnnoise_accepts_full_rt_block_without_short_write + the disabled twin, tests.rs:1887-1928
    (RT_PROCESS_BUFFER_CAPACITY = 8192, audio/processor.rs:47)."""
    p = oracle.RNNoiseProcessor()
    block = np.zeros(8192, dtype=np.float32)
    assert p.push_samples(block) == 8192
    p.process_frames()
    expected = (8192 // 480) * 480
    assert p.available_samples() == expected
    assert p.pending_input() == 8192 - expected
    assert p.read_samples(expected).size == expected
    p = oracle.RNNoiseProcessor()
    p.set_enabled(False)
    assert p.push_samples(block) == 8192
    p.process_frames()
    assert p.available_samples() == 8192 and p.pending_input() == 0


def test_ring_capacity_and_soft_reset(oracle):
    """rnnoise.rs:11 (capacity 8192 + 480), rt.rs:189-197 (push_slice accepts what fits), rnnoise.rs:216-232."""
    p = oracle.RNNoiseProcessor()
    assert p.push_samples(np.zeros(9000, dtype=np.float32)) == 8192 + 480
    assert p.push_samples(np.zeros(10, dtype=np.float32)) == 0
    p.soft_reset()
    assert p.pending_input() == 0 and p.available_samples() == 0
    # output ring full: process_frames stops while input frames remain (the `remaining() >= 480` condition)
    p = oracle.RNNoiseProcessor()
    accepted = []
    for _ in range(3):
        accepted.append(p.push_samples(np.zeros(8160, dtype=np.float32)))
        p.process_frames()
    assert accepted == [8160, 8160, 8672 - 7680]
    assert p.available_samples() == 18 * 480  # 8672 // 480 frames fit the output ring
    assert p.pending_input() == 8672


def test_streaming_in_ragged_blocks_equals_whole_frames(oracle):
    """push/process/pop in blocks of 1000 + 920 gives exactly what 1920 samples in one go give (rnnoise.rs:114-188)."""
    x = S.kat_signal(4)  # 1920 samples
    whole = oracle.suppressor_process(x, 1.0, 0x5EED)
    p = oracle.RNNoiseProcessor()
    got = []
    for lo, hi in ((0, 1000), (1000, 1920)):
        p.push_samples(x[lo:hi])
        p.process_frames()
        got.append(p.read_samples(p.available_samples()))
    assert [g.size for g in got] == [960, 960]
    assert np.array_equal(np.concatenate(got), whole)


def test_evaluation_order_independence(oracle):
    """The restatement's wavefront-native sums (order 0: what the GPU kernels evaluate) against the published scalar C's
    running sums (order 1).  Neither is "the" order of nnnoiseless (it unrolls over several accumulators), so the two
    must agree far inside the 1e-5 RMS budget and pick the same pitch, or the GPU == oracle tests would only prove
    self-consistency.  The count of differing pitch decisions is asserted, not hidden."""
    total_frames = 0
    flips = 0
    for index, gain in ((0, 1.0), (5, 2.0), (11, 0.2), (17, 0.02)):
        x = (S.kat_signal(250, *S.stream_params(index)) * np.float32(gain)).astype(np.float32)
        a, pa, sa = oracle.suppressor_process_traced(x, 1.0, 0x5EED, 0)
        b, pb, sb = oracle.suppressor_process_traced(x, 1.0, 0x5EED, 1)
        d = a.astype(np.float64) - b.astype(np.float64)
        assert float(np.sqrt(np.mean(d * d))) <= 1e-6, (index, gain)
        assert np.array_equal(sa, sb)
        total_frames += pa.size
        flips += int(np.count_nonzero(pa != pb))
    print(f"pitch decisions differing between evaluation orders: {flips} of {total_frames} frames")
    assert flips <= total_frames // 100


# ---------------------------------------------------------------- host-only part of the C ABI (no GPU work)
@pytest.fixture(scope="module")
def mi():
    import mic_eq_mi

    assert mic_eq_mi.CORE_AVAILABLE
    return mic_eq_mi


def test_noise_model_ids(mi):
    """test_noise_model_display_names / _from_id / test_available_models, noise_suppressor.rs:200-219."""
    M = mi.NoiseModel
    assert M.display_name(M.RNNOISE) == "RNNoise (Low Latency)"
    assert M.id(M.RNNOISE) == "rnnoise"
    assert M.from_id("rnnoise") == M.RNNOISE
    assert M.from_id("RNNOISE") == M.RNNOISE
    assert M.from_id("invalid") is None
    assert M.RNNOISE in M.available()
    # the deepfilter ids parse (noise_suppressor.rs:61-64) but the backend cannot be created here
    assert M.from_id("deepfilter-ll") == M.DEEPFILTER_LL and M.from_id("DeepFilterNet") == M.DEEPFILTER
    assert M.id(M.DEEPFILTER_LL) == "deepfilter-ll" and M.id(M.DEEPFILTER) == "deepfilter"
    for model in ("deepfilter-ll", "deepfilter"):
        with pytest.raises(NotImplementedError, match="DeepFilterNet backend is not built"):
            mi.new_noise_suppression_engine(model)
    with pytest.raises(ValueError):
        mi.NoiseSuppressor("no-such-model")


def test_weight_blob_layout(mi):
    """The fifteen int8 arrays of the RNNoise model, concatenated in declaration order, are what
    af_suppressor_load_weights takes: real weights drop in without any code change."""
    sizes = [42 * 24, 24, 24 * 72, 24 * 72, 72, 24 * 1, 1, 90 * 144, 48 * 144, 144, 114 * 288, 96 * 288, 288, 96 * 22, 22]
    assert len(sizes) == 15
    total = sum(sizes)
    assert total == 87_503
    eng = mi.Engine(48_000.0, 1)
    try:
        rng = np.random.default_rng(7)
        blob = rng.integers(-127, 128, size=total, dtype=np.int8).tobytes()
        eng.suppressor_load_weights(blob, total)
        with pytest.raises(ValueError, match="87503"):
            eng.suppressor_load_weights(blob[:-1], total - 1)
    finally:
        eng.close()
