"""tests/golden/{limiter_lookahead,dynamics_aliasing}.json were written by tools/gen_golden.py in the build container by
IMPORTING the reference's evaluators and running them with the CPU oracle behind `simulate_auto_eq_chain`.  Here (CPU only):

* this repository's own stimulus generators (tests/signals.py) reproduce the reference's stimuli bit for bit (SHA-256 over
  the float32 bytes, head, tail, 256 checkpoints);
* its own vectorised waveform metrics return what the reference's metric functions returned on the same oracle output;
* the reference harness over the oracle -- the fixture's `aggregate` / `rows` -- equals the published reports
  (`evaluation/limiter-lookahead-report.json`, `evaluation/dynamics-aliasing-report.json`): the oracle is pinned through the
  reference's own measuring code, including the waveform-level rows of the aliasing report.
"""
import numpy as np
import pytest

import signals as S


def test_own_generators_reproduce_the_reference_stimuli_bit_for_bit():
    for name, x in S.limiter_cases().items():
        assert x.dtype == np.float32
        assert S.fingerprint_mismatch(x, S.LIMITER_FIXTURE["stimuli"][name]) is None, name
    for name, carrier, modulation in S.ALIASING_CASES:
        for rate in (S.ALIASING_FIXTURE["base_rate"], S.ALIASING_FIXTURE["reference_rate"]):
            x = S.aliasing_signal(rate, carrier, modulation)
            assert S.fingerprint_mismatch(x, S.ALIASING_FIXTURE["stimuli"][name][str(rate)]) is None, (name, rate)


def test_settings_and_bands_are_the_captured_data():
    assert S.LIMITER_FIXTURE["lookahead_ms"] == [0.5, 1.0, 2.0]
    for lookahead in (0.5, 1.0, 2.0):
        settings = S.limiter_settings(lookahead)
        assert settings["limiter_lookahead_ms"] == lookahead and settings["limiter_careful_output_enabled"] is True
        assert settings == S.LIMITER_FIXTURE["settings"][f"{lookahead:g}"]
    assert len(S.LIMITER_BANDS) == 10 and len(S.ALIASING_BANDS) == 10
    assert S.ALIASING_SETTINGS["limiter_enabled"] is False and S.ALIASING_SETTINGS["compressor_ratio"] == 8.0


@pytest.mark.parametrize("lookahead_ms", [0.5, 1.0, 2.0])
def test_own_metrics_equal_the_reference_functions_on_the_oracle_output(oracle, lookahead_ms):
    rows = S.LIMITER_FIXTURE["rows"][f"{lookahead_ms:g}"]
    for name, x in S.limiter_cases().items():
        want = rows[name]
        r = oracle.simulate_auto_eq_chain(x, 48_000, S.LIMITER_BANDS, S.limiter_settings(lookahead_ms))
        out = np.asarray(r["output_audio"], dtype=np.float64)
        delay = want["declared_alignment_samples"]
        assert delay == int(round(lookahead_ms / 1000.0 * 48_000)) + 20
        aligned = out[delay:]
        ref = x[: aligned.size].astype(np.float64)
        picked = S.transient_indices(ref)
        assert picked.tolist() == want["transient_indices"], name
        assert len(picked) == want["transient_count"]
        assert abs(S.gain_envelope_variation_db(ref, aligned) - want["gain_envelope_variation_db"]) <= 1e-12, name
        assert abs(S.transient_error_db(ref, aligned, picked) - want["transient_shape_error_db"]) <= 1e-9, name
        assert r["true_peak_limited_events"] == want["true_peak_limited_events"]
        assert r["processed_samples"] == want["processed_samples"] and want["finite_output"] is True


# evaluation/limiter-lookahead-report.json, aggregates[ms]["controlled"] (the report's source hashes match the checkout)
PUBLISHED_LIMITER = {
    "2": {"median_gain_envelope_variation_db": 1.3907917598661823, "median_transient_shape_error_db": -44.837684744690314,
          "p90_transient_shape_error_db": -27.740017908805214, "max_true_peak_limiter_gain_reduction_db": 0.5307239890098572,
          "worst_pre_true_peak_overshoot_db": 0.5220339298248291, "total_true_peak_limited_events": 1},
    "0.5": {"median_gain_envelope_variation_db": 1.3906075587952735, "median_transient_shape_error_db": -44.837684744690314,
            "p90_transient_shape_error_db": -25.537891219961008},
}


def test_reference_harness_over_the_oracle_equals_the_published_limiter_report():
    for key, published in PUBLISHED_LIMITER.items():
        got = S.LIMITER_FIXTURE["aggregate"][key]
        assert got["cases"] == 3 and got["all_finite"] is True
        assert got["worst_output_true_peak_overshoot_db"] == 0.0 and got["minimum_main_peak_gain_reduction_db"] == 0.0
        for name, value in published.items():
            # waveform aggregates to the last digit; dB statistics that pass through an f32 log10 to 1 ulp of f32 (Windows CRT
            # vs glibc)
            tol = 2e-7 if name in ("max_true_peak_limiter_gain_reduction_db", "worst_pre_true_peak_overshoot_db") else 1e-12
            assert abs(got[name] - value) <= tol, (key, name, got[name], value)


# evaluation/dynamics-aliasing-report.json, cases[*]
PUBLISHED_ALIASING = {
    "carrier_8k": (-19.001835719980615, -43.454789994894405, 16.455915451049805, 17.137773513793945),
    "carrier_11k": (-34.82294332702055, -53.30385886557906, 16.86142921447754, 17.001239776611328),
    "carrier_15k": (-25.511439358287017, -47.70959094355128, 16.1809024810791, 16.64676856994629),
    "carrier_18k": (-19.869773892091406, -44.20170099571515, 15.484599113464355, 16.35672378540039),
}


def test_reference_harness_over_the_oracle_equals_the_published_aliasing_report():
    """Not only the peak gain reductions (round 1-2) but the report's waveform rows: the relative error between the 48 kHz
    render and the 192 kHz render, and the folded-product energy, to 1e-12 dB."""
    for name, (rel, folded, base_gr, ref_gr) in PUBLISHED_ALIASING.items():
        row = S.ALIASING_FIXTURE["rows"][name]
        assert row["alignment_lag_samples"] == 0
        assert abs(row["relative_waveform_error_db"] - rel) <= 1e-12, name
        assert abs(row["folded_out_of_expected_error_db"] - folded) <= 1e-12, name
        assert row["base_peak_gain_reduction_db"] == base_gr and row["reference_peak_gain_reduction_db"] == ref_gr, name


def test_own_aliasing_metrics_equal_the_reference_harness(oracle):
    for name, carrier, modulation in S.ALIASING_CASES[:2]:  # (two of the four: a 192 kHz oracle render takes a second)
        renders = []
        for rate in (48_000, 192_000):
            r = oracle.simulate_auto_eq_chain(S.aliasing_signal(rate, carrier, modulation), rate, S.ALIASING_BANDS, S.ALIASING_SETTINGS)
            renders.append(np.asarray(r["output_audio"], dtype=np.float64))
        got = S.aliasing_case_metrics(renders[0], renders[1], carrier, modulation)
        want = S.ALIASING_FIXTURE["rows"][name]
        assert got["alignment_lag_samples"] == want["alignment_lag_samples"]
        assert abs(got["relative_waveform_error_db"] - want["relative_waveform_error_db"]) <= 1e-9, name
        assert abs(got["folded_out_of_expected_error_db"] - want["folded_out_of_expected_error_db"]) <= 1e-9, name
