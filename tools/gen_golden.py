#!/usr/bin/env python3
"""Build-container-only: generate tests/golden/{limiter_lookahead,dynamics_aliasing}.json by IMPORTING the reference's own
evaluators from /root/reference/python (they import here: SURVEY.md 8(c)) and running them with this repository's CPU oracle
plugged in as `simulate_auto_eq_chain`.

What is written is data, never reference text:
  * the chain settings dicts and band lists the evaluators hand to `simulate_auto_eq_chain` (captured at the call);
  * fingerprints of their deterministic stimuli: SHA-256 over the float32 bytes, first / last 64 samples, 256 evenly spaced
    checkpoints, absolute sum -- tests/signals.py regenerates the stimuli in its own code and is held to these;
  * the outputs of the reference's metric functions (`_gain_envelope_variation_db`, `_transient_indices`,
    `_transient_error_db`, `_aggregate`; `_case` of the aliasing evaluator) on the ORACLE's output, so that the tests assert
    numbers and this repository's own metric code is checked against them.
The reference cannot travel to the GPU box in any form; nothing under tests/ or bench.py imports it.  Re-run:
    python tools/gen_golden.py            (needs /root/reference; writes tests/golden/*.json)
"""
from __future__ import annotations

import hashlib
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
REFERENCE = pathlib.Path("/root/reference")
GOLDEN = ROOT / "tests" / "golden"


def fingerprint(x: np.ndarray) -> dict:
    x = np.ascontiguousarray(x, dtype=np.float32)
    idx = np.linspace(0, x.size - 1, 256).astype(np.int64)
    return {
        "n": int(x.size),
        "sha256_f32le": hashlib.sha256(x.astype("<f4").tobytes()).hexdigest(),
        "head": [float(v) for v in x[:64]],
        "tail": [float(v) for v in x[-64:]],
        "checkpoints": [[int(i), float(x[i])] for i in idx],
        "abs_sum_f64": float(np.sum(np.abs(x.astype(np.float64)))),
        "nonzero": int(np.count_nonzero(x)),
    }


def file_sha256(path: pathlib.Path) -> str:
    return hashlib.sha256(path.read_bytes()).hexdigest()


def main() -> None:
    if not REFERENCE.exists():
        raise SystemExit("tools/gen_golden.py runs in the build container only (/root/reference is not here)")
    for p in (ROOT / "oracle", REFERENCE / "python", REFERENCE / "python" / "tools"):
        sys.path.insert(0, str(p))
    try:  # the evaluators import `tomllib` (Python 3.11); this image has 3.10 and the same parser as `tomli`
        import tomllib  # noqa: F401
    except ImportError:
        import tomli

        sys.modules["tomllib"] = tomli
    import af_oracle_py as oracle
    import evaluate_dynamics_aliasing as A
    import evaluate_limiter_lookahead as L

    calls = []

    def chain(audio, sample_rate, bands, settings):  # the oracle behind the reference's operator name
        calls.append((int(sample_rate), [list(map(float, b)) for b in bands], dict(settings)))
        r = dict(oracle.simulate_auto_eq_chain(np.ascontiguousarray(audio, dtype=np.float32), sample_rate, bands, settings))
        r.setdefault("candidate_runtime_ms", 0.0)
        return r

    # ---------------------------------------------------------------- limiter lookahead evaluator
    L.simulate_auto_eq_chain = chain
    cases = L._cases()
    out = {
        "generated_by": "tools/gen_golden.py",
        "reference_file": "python/tools/evaluate_limiter_lookahead.py",
        "reference_file_sha256": file_sha256(REFERENCE / "python/tools/evaluate_limiter_lookahead.py"),
        "sample_rate": int(L.SAMPLE_RATE),
        "lookahead_ms": [float(v) for v in L.LOOKAHEAD_MS],
        "stimuli": {name: fingerprint(x) for name, x in cases.items()},
        "settings": {},
        "rows": {},
        "aggregate": {},
    }
    for lookahead in L.LOOKAHEAD_MS:
        calls.clear()
        rows = [L._case(name, audio, lookahead) for name, audio in cases.items()]
        fs, bands, settings = calls[-1]
        out["bands"] = bands
        out["settings"][f"{lookahead:g}"] = settings
        keep = ("id", "lookahead_ms", "pre_true_peak_overshoot_db", "output_true_peak_overshoot_db", "main_peak_gain_reduction_db",
                "true_peak_limiter_gain_reduction_db", "true_peak_limited_events", "gain_envelope_variation_db",
                "transient_shape_error_db", "transient_count", "finite_output", "processed_samples", "declared_alignment_samples")
        per_case = {}
        for row, (name, audio) in zip(rows, cases.items()):
            item = {k: row[k] for k in keep}
            # the intermediate the metric functions work on, so that this repository's own metric code can be checked stage by stage
            rendered = np.asarray(chain(audio, fs, bands, settings)["output_audio"], dtype=np.float64)
            aligned = rendered[row["declared_alignment_samples"]:]
            reference = audio[: aligned.size].astype(np.float64)
            item["transient_indices"] = [int(v) for v in L._transient_indices(reference)]
            assert item["gain_envelope_variation_db"] == L._gain_envelope_variation_db(reference, aligned)
            per_case[name] = item
        out["rows"][f"{lookahead:g}"] = per_case
        agg = L._aggregate(rows)
        out["aggregate"][f"{lookahead:g}"] = {k: v for k, v in agg.items() if "runtime" not in k}
    published = json.loads((REFERENCE / "evaluation" / "limiter-lookahead-report.json").read_text())
    out["published_report_sha256"] = file_sha256(REFERENCE / "evaluation" / "limiter-lookahead-report.json")
    (GOLDEN / "limiter_lookahead.json").write_text(json.dumps(out, indent=1) + "\n")
    print("limiter_lookahead.json:", {k: out["aggregate"][k].get("median_gain_envelope_variation_db") for k in out["aggregate"]})
    del published

    # ---------------------------------------------------------------- dynamics aliasing evaluator
    A.simulate_auto_eq_chain = chain
    out = {
        "generated_by": "tools/gen_golden.py",
        "reference_file": "python/tools/evaluate_dynamics_aliasing.py",
        "reference_file_sha256": file_sha256(REFERENCE / "python/tools/evaluate_dynamics_aliasing.py"),
        "base_rate": int(A.BASE_RATE),
        "reference_rate": int(A.REFERENCE_RATE),
        "cases": [[name, float(c), float(m)] for name, c, m in A.CASES],
        "stimuli": {},
        "rows": {},
    }
    for name, carrier, modulation in A.CASES:
        calls.clear()
        row = A._case(name, carrier, modulation)
        out["rows"][name] = {k: v for k, v in row.items() if "runtime" not in k}
        out["stimuli"][name] = {str(rate): fingerprint(A._signal(rate, carrier, modulation)) for rate in (A.BASE_RATE, A.REFERENCE_RATE)}
        out["bands"] = calls[-1][1]
        out["settings"] = calls[-1][2]
    (GOLDEN / "dynamics_aliasing.json").write_text(json.dumps(out, indent=1) + "\n")
    print("dynamics_aliasing.json:", {k: v.get("base_peak_gain_reduction_db") for k, v in out["rows"].items()})


if __name__ == "__main__":
    main()
