"""Command-line twin of the reference's `rnnoise_benchmark` binary (rust-core/src/bin/rnnoise_benchmark.rs:51-117):

    python -m mic_eq_mi.rnnoise_benchmark <input.f32> <output.f32> <metadata.json> [--weights blob.i8]

Raw little-endian f32 in, raw little-endian f32 out, JSON metadata with the reference's keys.  The evaluation
harness takes such an executable through `evaluate_rnnoise_backends.py --shipped-binary`.
"""
from __future__ import annotations

import sys


def main(argv: list[str]) -> int:
    args = [a for a in argv if not a.startswith("--")]
    if len(args) != 3:
        print("usage: rnnoise_benchmark <input.f32> <output.f32> <metadata.json>", file=sys.stderr)  # rnnoise_benchmark.rs:55-60
        return 2
    from . import mic_eq_core

    mic_eq_core.rnnoise_benchmark(args[0], args[1], args[2])
    return 0


if __name__ == "__main__":
    raise SystemExit(main(sys.argv[1:]))
