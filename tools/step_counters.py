#!/usr/bin/env python3
"""Fold the per-group counter summaries of tools/step_counters.sh (gpurun_out/stepc_<n>.json) into
gpurun_out/<round>_step_counters[_<shape>].json: per kernel and per bench STEP (the runs hold 1 warm-up + 1 timed step): launches, VALU
wave-instructions, f64 flops, HBM fetch / write bytes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KB;
on gfx950 FETCH_SIZE counts a 16-B-per-lane streaming read at half its bytes -- the chain kernel's float4 chunk loads --
and is doubled for that kernel only; the suppressor's kernels read 4 and 8 B per lane, for which the counter is taken as
is and marked uncalibrated), SQ activity."""
import json
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
STEPS_IN_RUN = 2  # --warmup 1 --steps 1
args = sys.argv[1:]
chain = "dynamics" if "dynamics" in args else "full"
merged: dict = {}
for p in range(1, 6):
    path = ROOT / "gpurun_out" / f"stepc_{p}.json"
    if not path.exists():
        continue
    for name, counters in json.loads(path.read_text()).items():
        row = merged.setdefault(name, {})
        for counter, v in counters.items():
            row[counter] = {"dispatches": v["dispatches"], "total": v["mean_per_dispatch"] * v["dispatches"]}
durations = {}  # kernel name -> (average ms per launch, launches) from the counter-free pass
stats_path = ROOT / "gpurun_out" / "stepc_kernel_stats.csv"
if stats_path.exists():
    import csv

    for r in csv.DictReader(open(stats_path)):
        durations[r["Name"]] = (float(r["AverageNs"]) / 1e6, int(r["Calls"]))
kernels = {}
for name, row in merged.items():
    def per_step(counter):
        return row[counter]["total"] / STEPS_IN_RUN if counter in row else None
    launches = max((v["dispatches"] for v in row.values()), default=0) / STEPS_IN_RUN
    fetch_kb, write_kb = per_step("FETCH_SIZE"), per_step("WRITE_SIZE")
    wide = "chain_ring" in name or "chain_quad" in name
    f64 = None
    if "SQ_INSTS_VALU_FMA_F64" in row:
        f64 = 64.0 * (per_step("SQ_INSTS_VALU_ADD_F64") + per_step("SQ_INSTS_VALU_MUL_F64") + per_step("SQ_INSTS_VALU_TRANS_F64")
                      + 2.0 * per_step("SQ_INSTS_VALU_FMA_F64"))
    kernels[name] = {
        "launches_per_step": launches,
        "kernel_ms_per_launch": durations.get(name, (None, 0))[0],  # rocprofv3 --kernel-trace --stats, own run of the same command
        "fetch_bytes": (fetch_kb or 0.0) * 1024.0 * (2.0 if wide else 1.0),
        "fetch_correction": "x2 (16 B per lane streaming reads)" if wide else "none (4/8 B per lane reads: uncalibrated)",
        "write_bytes": (write_kb or 0.0) * 1024.0,
        "valu_insts": per_step("SQ_INSTS_VALU") or 0.0,
        "f64_flops": f64 or 0.0,
        "sq": {c: per_step(c) for c in ("SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                        "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_INSTS_SALU",
                                        "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA",
                                        "SQ_INSTS_VALU_MFMA_MOPS_F64", "GRBM_GUI_ACTIVE") if c in row},
    }
try:
    commit = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    commit = ""
streams, seconds = 4096, 10.0
for i, a in enumerate(args):
    if a == "--streams":
        streams = int(args[i + 1])
    if a == "--seconds":
        seconds = float(args[i + 1])
out = {
    "_how": "tools/step_counters.sh: five separate `rocprofv3 --pmc <group> --kernel-trace` runs of `python bench.py --steps 1 --warmup 1 "
            "--no-cpu-baseline` (FETCH_SIZE | WRITE_SIZE | SQ instruction counts | SQ activity | matrix-core counters), per-kernel totals divided by the 2 steps of a run",
    "commit": commit or None, "shape": {"streams": streams, "seconds": seconds, "chain": chain, "auto_makeup": "--auto-makeup" in args, "deesser": "--deesser" in args},
    "kernels": kernels,
}
import os

tag = os.environ.get("ROUND", "r03")
suffix = "" if (chain, streams, "--auto-makeup" in args, "--deesser" in args) == ("full", 4096, False, False) else (
    f"_{chain}_{streams}" + ("_automakeup" if "--auto-makeup" in args else "") + ("_deesser" if "--deesser" in args else ""))
dest = ROOT / "gpurun_out" / f"{tag}_step_counters{suffix}.json"
dest.write_text(json.dumps(out, indent=1, sort_keys=True))
total = sum(k["fetch_bytes"] + k["write_bytes"] for k in kernels.values())
print(f"wrote {dest}: {len(kernels)} kernels, HBM traffic per step {total / 1e9:.2f} GB")
for name, k in sorted(kernels.items(), key=lambda kv: -(kv[1]['fetch_bytes'] + kv[1]['write_bytes'])):
    sq = k["sq"]
    conflict = (sq.get("SQ_LDS_BANK_CONFLICT") or 0.0) / max(sq.get("SQ_ACTIVE_INST_LDS") or 1.0, 1.0)
    print(f"  {name[:44]:44s} launches {k['launches_per_step']:5.1f} fetch {k['fetch_bytes'] / 1e9:7.2f} GB write {k['write_bytes'] / 1e9:7.2f} GB "
          f"valu {k['valu_insts']:.3e} f64 {k['f64_flops']:.3e} flop  LDS conflict/active {conflict:.2f}")
