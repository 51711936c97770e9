"""measure_integrated_loudness (lib.rs:290-298, loudness.rs:43-83) on the GPU against the CPU oracle, plus the
reference's own test properties (loudness.rs:222-257).  ebur128 itself is not vendored: both sides restate the
published gating design in histogram mode (parity with the crate unpinned)."""
import numpy as np
import pytest

import signals as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mi():
    import mic_eq_mi

    assert mic_eq_mi.CORE_AVAILABLE, "HIP library missing: GPU tests never fall back to the CPU"
    return mic_eq_mi


def _tone(seconds, amp=0.1, fs=48_000):
    i = np.arange(int(fs * seconds), dtype=np.float32)
    return (np.float32(amp) * np.sin(np.float32(2.0 * np.pi) * np.float32(1000.0) * i / np.float32(fs))).astype(np.float32)


def test_matches_oracle_exactly(mi, oracle):
    from mic_eq_mi import mic_eq_core as core

    clips = [S.kat_signal(300, *S.stream_params(s)) for s in range(67)]
    batch = np.stack(clips)
    batch[5] *= np.float32(0.01)
    lufs, status = core.measure_integrated_loudness_batch(batch, 48_000)
    assert not status.any()
    for s in (0, 5, 63, 64, 66):
        assert lufs[s] == oracle.measure_integrated_loudness(batch[s], 48_000), s
    # single-clip operator, other rates, ragged length
    for fs in (44_100, 16_000, 96_000):
        x = _tone(3.37, fs=fs)
        assert mi.measure_integrated_loudness(x, fs) == oracle.measure_integrated_loudness(x, fs)


def test_reference_properties(mi):
    """loudness.rs:222-257."""
    tone = _tone(8.0)
    padded = np.concatenate([np.zeros(48_000, np.float32), tone, np.zeros(48_000, np.float32)])
    a, b = mi.measure_integrated_loudness(tone, 48_000), mi.measure_integrated_loudness(padded, 48_000)
    assert abs(a - b) < 0.2
    assert -24.0 < a < -22.0  # -20 dBFS 1 kHz sine: -23.0 LUFS
    with pytest.raises(ValueError):
        mi.measure_integrated_loudness(np.zeros(0, np.float32), 48_000)
    with pytest.raises(ValueError):
        mi.measure_integrated_loudness(np.array([np.nan], np.float32), 48_000)
    with pytest.raises(ValueError):
        mi.measure_integrated_loudness(np.array([0.1], np.float32), 12_345)
    with pytest.raises(ValueError):  # silence never passes the absolute gate
        mi.measure_integrated_loudness(np.zeros(48_000, np.float32), 48_000)
