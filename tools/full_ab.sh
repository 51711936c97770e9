#!/bin/bash
# Same-box A/B of the default bench step under environment switches: tools/full_ab.sh name:VAR=val[,VAR=val] ...
# (BENCH_ARGS="--auto-makeup" adds arguments to every run)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%:*}"; vars="${spec#*:}"
  env ${vars//,/ } python bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > "gpurun_out/full_${name}.json" 2> "gpurun_out/full_${name}.err"
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
l = json.loads([x for x in open(f"gpurun_out/full_{n}.json") if x.startswith("{")][-1])
print(f"{n}: {l['ms_per_step']:.1f} ms/step  supp {l['stage_ms']['suppressor_and_front_end']:.1f}  chain(sum) {l['stage_ms']['chain']:.1f}  rms {l['checks']['output_rms']:.9f}")
PY
done
