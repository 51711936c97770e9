"""Kernel 5, the role pipeline (csrc/af_roles.hip: compressor and limiter as two kernels of dedicated serial waves and
feed-forward waves, LDS hand-over) against the token-ring kernel: the expressions are the same operation for operation, so
audio, every block row and the state the kernels leave behind must agree BIT FOR BIT -- including a stream that changes
kernel between calls.  (The token-ring kernel itself is held to the CPU oracle by tests/test_gpu_parity.py; the variant
`roles` of that file runs the oracle comparisons on this kernel directly.)"""
import numpy as np
import pytest

import signals as S

pytestmark = pytest.mark.gpu

ROLES, RING = 5, 2


def _run(kernels, audio, cuts, settings, bands=None, fs=48_000.0, suppressor=False, control_block=None):
    """One engine over `audio` cut into calls at `cuts`; call i runs kernel kernels[i % len(kernels)]."""
    from mic_eq_mi import mic_eq_core as core

    eng = core.Engine(fs, audio.shape[0])
    core.configure_auto_eq_chain(eng, fs, bands or S.LIMITER_BANDS, settings)
    if control_block:
        eng.set_control_block_samples(control_block)
    if suppressor:
        eng.set_prefilter_enabled(1, 1)
        eng.set_suppressor_enabled(1)
    outs, rows, used = [], [], []
    edges = [0, *cuts, audio.shape[1]]
    for i, (lo, hi) in enumerate(zip(edges[:-1], edges[1:])):
        eng.set_kernel(kernels[i % len(kernels)])
        outs.append(eng.process(audio[:, lo:hi]))
        rows.append(eng.block_stats().copy())
        used.append(eng.last_kernel())
    eng.close()
    return np.concatenate(outs, axis=1), np.concatenate(rows, axis=0), used


def _same(a, b):
    assert a[0].shape == b[0].shape
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), f"audio differs: max abs {np.abs(a[0] - b[0]).max():.3e}"
    for name in a[1].dtype.names:
        assert a[1][name].tobytes() == b[1][name].tobytes(), f"block rows differ in {name}"


CONFIGS = {
    "bench-chain": dict(S.limiter_settings(2.0)),
    "adaptive-release": dict(S.limiter_settings(1.0), compressor_adaptive_release=True, compressor_threshold_db=-26.0, compressor_ratio=2.5),
    "no-sidechain-makeup": dict(S.limiter_settings(0.5), compressor_sidechain_highpass_enabled=False, compressor_makeup_gain_db=6.0,
                                compressor_attack_ms=2.0),
    "limiter-only": dict(S.limiter_settings(2.0), compressor_enabled=False),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_roles_equal_the_token_ring_bit_for_bit(name):
    settings = CONFIGS[name]
    audio = (S.batch_signal(70, 9) * np.float32(8.0)).astype(np.float32)  # 70 streams (64 + 6), 4320 samples, hot enough to limit
    bands = [(80.0 * 1.75**i, 3.0 if i % 2 else -2.5, 1.0) for i in range(10)]  # legacy setters: a crossfade opens the stream
    cuts = [1920, 3003]  # whole control blocks, then a call that ends inside a block and inside a tile
    ring = _run([RING], audio, cuts, settings, bands)
    roles = _run([ROLES], audio, cuts, settings, bands)
    assert set(ring[2]) == {RING} and set(roles[2]) == {ROLES}
    _same(ring, roles)
    # the kernels share the state planes: a stream may change kernel between calls
    mixed = _run([ROLES, RING, ROLES], audio, cuts, settings, bands)
    assert mixed[2] == [ROLES, RING, ROLES]
    _same(ring, mixed)
    # both limiters really worked on this input
    assert float(ring[1]["limiter_peak_gain_reduction_db"].max()) > 0.5 and int(ring[1]["true_peak_limited_events"].sum()) > 0


def test_roles_behind_the_suppressor_equal_the_token_ring():
    """Full chain (front end + RNNoise suppressor + systolic EQ + dynamics), several suppressor windows over two calls."""
    audio = S.batch_signal(70, 150)
    settings = S.limiter_settings(2.0)
    ring = _run([RING], audio, [90 * 480], settings, suppressor=True)
    roles = _run([ROLES], audio, [90 * 480], settings, suppressor=True)
    assert set(roles[2]) == {ROLES}
    _same(ring, roles)


def test_configurations_the_roles_do_not_build_fall_back():
    """Auto-makeup (and the de-esser) stay on the token ring; asking for kernel 5 then runs the ring and says so."""
    audio = S.batch_signal(3, 8)
    settings = dict(S.limiter_settings(2.0), compressor_auto_makeup_enabled=True)
    roles = _run([ROLES], audio, [], settings)
    ring = _run([RING], audio, [], settings)
    assert roles[2] == [RING]
    _same(ring, roles)
