#!/usr/bin/env python3
"""How the resampler's f32 cutoffs in oracle/af_resampler.c (and csrc/af_resampler_host.hpp) were identified.

rubato 0.14.1's `calculate_cutoff(sinc_len, window)` is not readable offline.  For each configuration the
reference measured (tests/golden/resampler_report_pins.json) this script bisects k in
f_cutoff = f32(1 / (1 + k / sinc_len)) until ONE published figure is reproduced, then prints the f32
neighbours (they miss in the 5th digit) so that the identification is visibly unique.  The other
published figures are checked by tests/test_oracle_resampler.py with no freedom left.

    python tools/fit_resampler_cutoff.py
"""
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "oracle"), str(ROOT / "tests")]
import af_oracle_py as oracle  # noqa: E402
import resampler_stimuli as R  # noqa: E402

PINS = json.loads((ROOT / "tests" / "golden" / "resampler_report_pins.json").read_text())


def run(x, fi, fo, fc, sinc_len, window):
    y, _, expected, _ = oracle.simulate_product_resampler(x, fi, fo, 1024, sinc_len, window, fc)
    return y[:expected]


def fc_of(k, n):
    return float(np.float32(1.0 / (1.0 + k / n)))


def bisect(measure, target, k_lo, k_hi, n):
    rising = measure(fc_of(k_hi, n)) > measure(fc_of(k_lo, n))
    for _ in range(48):
        mid = 0.5 * (k_lo + k_hi)
        if (measure(fc_of(mid, n)) < target) == rising:
            k_lo = mid
        else:
            k_hi = mid
    return k_lo


def report(name, measure, target, k_lo, k_hi, n):
    k = bisect(measure, target, k_lo, k_hi, n)
    fc = np.float32(fc_of(k, n))
    print(f"{name}: k = {k:.6f}  f_cutoff = {float(fc)!r} (0x{fc.view(np.uint32):08X})  -> {measure(float(fc))!r}  published {target!r}")
    for step, toward in ((-1, 0.0), (1, 2.0)):
        nb = np.nextafter(fc, np.float32(toward))
        print(f"    neighbour {step:+d} ulp {float(nb)!r} -> {measure(float(nb))!r}")


def main():
    noise = R.stopband_noise()
    ref_rms = R.rms(R.steady(noise, 48_000))
    report("blackman/128 (product)",
           lambda fc: R.db_ratio(R.rms(R.steady(run(noise, 48_000, 44_100, fc, 128, "blackman"), 44_100)), ref_rms),
           PINS["product"]["swept_noise_attenuation_db"], 6.0, 6.6, 128)

    def edge_gain(fi, fo, sinc_len, window):
        s = R.sine(fi, 20_000.0, 1.5)
        den = R.rms(R.steady(s, fi))
        return lambda fc: R.db_ratio(R.rms(R.steady(run(s, fi, fo, fc, sinc_len, window), fo)), den)

    report("blackman_harris_squared/128",
           edge_gain(44_100, 48_000, 128, "blackman_harris_squared"),
           -PINS["legacy-blackman-harris-squared-128"]["passband_max_absolute_error_db"]["44100->48000"], 13.0, 20.0, 128)
    report("blackman_harris_squared/256",
           edge_gain(48_000, 44_100, 256, "blackman_harris_squared"),
           -PINS["high-rejection-blackman-harris-squared-256"]["passband_max_absolute_error_db"]["48000->44100"], 14.0, 15.0, 256)


if __name__ == "__main__":
    main()
