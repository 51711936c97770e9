#!/bin/bash
# Counter passes of the default bench step (one rocprofv3 --pmc run per counter group, kernel trace only), folded into
# gpurun_out/${ROUND:-r03}_step_counters.json by tools/step_counters.py (copy it to profiles/ to have bench.py read it).
#   tools/step_counters.sh [bench args, default: full chain]
cd "$(dirname "$0")/.."
root="$PWD"
mkdir -p gpurun_out
( cd /tmp && rocprofv3 -L > "$root/gpurun_out/${ROUND:-r03}_counter_list.txt" 2>&1 || true )
# pass 0: kernel durations of the same command without counters (rocprofv3 --kernel-trace --stats): the per-kernel averages the
# counter figures are divided by, and the summary that goes to profiles/ as <round>_..._kernel_stats.csv
rm -rf gpurun_out/stepc_0
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/stepc_0" -- python "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$root/gpurun_out/stepc_0.log" 2>&1 || true )
python - <<'PY'
import csv, glob
rows = []
for path in glob.glob("gpurun_out/stepc_0/**/*kernel_stats.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(path)) if any(t in r["Name"] for t in ("supp_", "chain_", "eq_systolic", "eq_stream", "stage_", "deesser", "resample"))]
if rows:
    with open("gpurun_out/stepc_kernel_stats.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(sorted(rows, key=lambda r: -float(r["TotalDurationNs"])))
PY
rm -rf gpurun_out/stepc_0
pass=0
for counters in "FETCH_SIZE" "WRITE_SIZE" \
                "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
                "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32" \
                "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"; do
  pass=$((pass + 1))
  rm -rf "gpurun_out/stepc_${pass}"
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$root/gpurun_out/stepc_${pass}" -- python "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline "$@" > "$root/gpurun_out/stepc_${pass}.log" 2>&1 || true )
  python tools/pmc_summary.py "gpurun_out/stepc_${pass}" "gpurun_out/stepc_${pass}.json" supp_ chain_ eq_systolic eq_stream stage_ > /dev/null || echo "pass ${pass} (${counters}) produced no counters"
  rm -rf "gpurun_out/stepc_${pass}"
done
python tools/step_counters.py "$@"
