"""Noise gate (dsp/gate.rs, the expander path simulate_gate_suppressor_order exercises) on the GPU against the CPU
oracle, the reference's own gate tests (gate.rs:958-1103) on the GPU output, and the operator in both orders."""
import numpy as np
import pytest

import signals as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mi():
    import mic_eq_mi

    assert mic_eq_mi.CORE_AVAILABLE, "HIP library missing: GPU tests never fall back to the CPU"
    return mic_eq_mi


def _err(a, b):
    d = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(d))), float(np.sqrt(np.mean(d * d)))


def test_gate_matches_oracle(mi, oracle):
    """sqrt / log10 / exp10 per sample: device vs glibc differ by <= 2 ulp of f64 -> max |err| <= 2e-7 on the audio."""
    from mic_eq_mi import mic_eq_core as core

    n_streams = 67
    audio = np.stack([S.kat_signal(150, *S.stream_params(s)) * np.float32(0.02 + 0.03 * (s % 5)) for s in range(n_streams)])
    audio[:, 20_000:30_000] *= np.float32(0.01)  # a pause: the expander closes, then reopens
    out, trace, chatter = core.gate_batch(audio, -40.0, 10.0, 100.0, 48_000.0, True, 480)
    for s in (0, 1, 63, 64, 66):
        g = oracle.Gate(-40.0, 10.0, 100.0, 48_000.0, vad_mode=True)
        want = np.empty_like(audio[s])
        gains = []
        for f in range(audio.shape[1] // 480):
            want[f * 480 : (f + 1) * 480] = g.process(audio[s, f * 480 : (f + 1) * 480])
            gains.append(g.current_gain)
        max_abs, rms = _err(out[s], want)
        assert max_abs <= 2e-7 and rms <= 2e-8, (s, max_abs, rms)
        assert np.max(np.abs(trace[:, s] - np.asarray(gains, dtype=np.float32))) <= 1e-6
        assert int(chatter[s]) == g.chatter_event_count
    assert trace.min() < 0.2 < 0.8 < trace.max()  # the gate really closed and opened


def test_reference_gate_properties(mi):
    """gate.rs:958-1070 driven through the GPU kernel (one stream per scenario)."""
    from mic_eq_mi import mic_eq_core as core

    def run(x, attack, release, block=None):
        x = np.asarray(x, dtype=np.float32).reshape(1, -1)
        return core.gate_batch(x, -40.0, attack, release, 48_000.0, False, block or x.shape[1])

    _, tr, _ = run(np.full(3_000, 0.1), 10.0, 100.0)
    assert tr[-1, 0] > 0.8                                            # opens above threshold
    _, tr, _ = run(np.concatenate([np.full(3_000, 0.1), np.full(10_000, 0.0001)]), 10.0, 100.0, 1_000)
    assert tr[-1, 0] < tr[2, 0] * 0.7 and tr[-1, 0] < 0.5            # closes below threshold
    _, tr, _ = run(np.zeros(4_000), 1.0, 1.0)
    assert abs(tr[-1, 0] - 10.0 ** (-36.0 / 20.0)) < 0.02             # range cap
    _, hi, _ = run(np.full(2_000, 0.1), 1.0, 1.0)
    _, lo, _ = run(np.full(2_000, 0.0005), 1.0, 1.0)
    assert hi[-1, 0] > lo[-1, 0]                                      # monotonic expander
    burst = np.concatenate([np.concatenate([np.full(2_000, 0.1), np.zeros(4_500)]) for _ in range(5)])
    _, _, chatter = run(burst, 1.0, 10.0)
    assert chatter[0] > 0                                             # rapid chatter detected


@pytest.mark.parametrize("suppressor_before_gate", [True, False])
def test_operator_both_orders(mi, oracle, suppressor_before_gate):
    x = (S.kat_signal(61) * np.float32(0.2))[: 60 * 480 + 123]      # ragged: last frame zero padded
    x[9_000:16_000] *= np.float32(0.005)
    probs = [0.5] * 61
    got = mi.simulate_gate_suppressor_order(x, probs, suppressor_before_gate, 0.8, {"gate_release_ms": 60.0})
    want = oracle.simulate_gate_suppressor_order(x, probs, suppressor_before_gate, 0.8, {"gate_release_ms": 60.0})
    assert set(got) == {"output_audio", "gate_gain", "gate_chatter_event_count", "gate_noise_floor_db",
                        "gate_noise_floor_reliability", "suppressor_latency_samples", "runtime_ms"}
    assert len(got["output_audio"]) == x.size and len(got["gate_gain"]) == 61
    d = np.asarray(got["output_audio"], dtype=np.float64) - want["output_audio"].astype(np.float64)
    assert float(np.sqrt(np.mean(d * d))) <= 1e-5
    assert np.max(np.abs(np.asarray(got["gate_gain"]) - np.asarray(want["gate_gain"]))) <= 1e-4
    assert got["gate_chatter_event_count"] == want["gate_chatter_event_count"]
    assert (got["gate_noise_floor_db"], got["gate_noise_floor_reliability"], got["suppressor_latency_samples"]) == (-60.0, 0.0, 480)
    with pytest.raises(ValueError):
        mi.simulate_gate_suppressor_order(x, probs[:-1], suppressor_before_gate)
    with pytest.raises(ValueError):
        mi.simulate_gate_suppressor_order(x, probs, suppressor_before_gate, 1.5)
    with pytest.raises(ValueError):
        mi.simulate_gate_suppressor_order(x, [2.0] * 61, suppressor_before_gate)
