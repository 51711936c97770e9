#!/bin/bash
# Instruction-cache, scalar-cache and scratch / vector-memory counters of the chain kernel inside the default bench step
# (rocprofv3 --pmc serialises dispatches: these are the kernel's own figures, without the suppressor beside it).
cd "$(dirname "$0")/.."
root="$PWD"
mkdir -p gpurun_out
pass=0
for counters in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES" \
                "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES" \
                "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD"; do
  pass=$((pass + 1))
  rm -rf "gpurun_out/ccp_${pass}"
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$root/gpurun_out/ccp_${pass}" -- python "$root/bench.py" --steps 1 --warmup 1 --no-cpu-baseline "$@" > "$root/gpurun_out/ccp_${pass}.log" 2>&1 || true )
  python tools/pmc_summary.py "gpurun_out/ccp_${pass}" "gpurun_out/ccp_${pass}.json" chain_ > /dev/null || echo "pass ${pass} failed"
  rm -rf "gpurun_out/ccp_${pass}"
done
python - <<'PY'
import json
for p in (1, 2, 3):
    try:
        d = json.load(open(f"gpurun_out/ccp_{p}.json"))
    except Exception as e:
        print(p, "no file", e); continue
    for k, v in d.items():
        print(k[:60], {c: round(x["mean_per_dispatch"]) for c, x in v.items()})
PY
