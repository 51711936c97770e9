#!/bin/bash
# Same-box A/B of chain-kernel builds: tools/ab_chain.sh <name>=<lib.so>[:variant] ...  -> one line per build (dynamics
# chain, 4096 x 10 s); libraries come from `make OUT=... OBJDIR=... EXTRA=...` in audio-forge_amd/csrc.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%=*}"; rest="${spec#*=}"; lib="${rest%%:*}"; variant=""
  if [[ "$rest" == *:* ]]; then variant="${rest#*:}"; fi
  AF_LIB_PATH="$PWD/$lib" python bench.py --chain dynamics --steps 3 --warmup 1 --no-cpu-baseline ${variant:+--variant $variant} > "gpurun_out/ab_${name}.json" 2> "gpurun_out/ab_${name}.err" || { tail -5 "gpurun_out/ab_${name}.err"; exit 1; }
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
line = json.loads([l for l in open(f"gpurun_out/ab_{name}.json") if l.startswith("{")][-1])
print(f"{name}: {line['ms_per_step']:.1f} ms/step, chain kernel {line['roofline']['avg_kernel_ms']:.2f} ms x {line['roofline']['launches_per_step']}, checks {line['checks']['output_rms']:.9f} {line['config']['kernel']}")
PY
done
