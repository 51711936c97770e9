#!/bin/bash
# Counter passes of the default bench step (one rocprofv3 --pmc run per counter group, kernel trace only), folded into
# profiles/r02_step_counters.json by tools/step_counters.py.   tools/step_counters.sh [bench args, default: full chain]
cd "$(dirname "$0")/.."
root="$PWD"
mkdir -p gpurun_out
( cd /tmp && rocprofv3 -L > "$root/gpurun_out/r02_counter_list.txt" 2>&1 || true )
pass=0
for counters in "FETCH_SIZE" "WRITE_SIZE" \
                "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
                "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  pass=$((pass + 1))
  rm -rf "gpurun_out/stepc_${pass}"
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$root/gpurun_out/stepc_${pass}" -- python "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline "$@" > "$root/gpurun_out/stepc_${pass}.log" 2>&1 || true )
  python tools/pmc_summary.py "gpurun_out/stepc_${pass}" "gpurun_out/stepc_${pass}.json" supp_ chain_ eq_systolic stage_ > /dev/null || echo "pass ${pass} (${counters}) produced no counters"
  rm -rf "gpurun_out/stepc_${pass}"
done
python tools/step_counters.py "$@"
