#!/bin/bash
# Suppressor-only measurements on one box: tools/supp_profile.sh <tag> [lib.so]
#   1. span of the suppressor's pipeline without the chain (bench workload, AF_DIAG_SKIP_CHAIN=1)
#   2. per-kernel times without overlap (AF_SERIAL_STREAMS=1, uniform 50-frame windows) from rocprofv3 --kernel-trace --stats
set -e
cd "$(dirname "$0")/.."
tag="$1"; lib="${2:-audio-forge_amd/libaudioforge_mi.so}"
mkdir -p gpurun_out
export AF_LIB_PATH="$PWD/$lib"
AF_DIAG_SKIP_CHAIN=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "gpurun_out/supp_${tag}_span.json" 2> "gpurun_out/supp_${tag}_span.err"
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
line = json.loads([l for l in open(f"gpurun_out/supp_{tag}_span.json") if l.startswith("{")][-1])
print(f"{tag}: suppressor span {line['stage_ms']['suppressor_and_front_end']:.1f} ms per step (all kernels {line['stage_ms']['all_kernels']:.1f})")
PY
rm -rf "gpurun_out/prof_supp_${tag}"
( cd /tmp && export TMPDIR=/tmp && AF_SERIAL_STREAMS=1 AF_SUPP_RAMP=0 AF_SUPP_WINDOW_FRAMES=50 AF_DIAG_SKIP_CHAIN=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_supp_${tag}" -- python "$OLDPWD/bench.py" --steps 1 --warmup 1 --seconds 2 --no-cpu-baseline > "$OLDPWD/gpurun_out/supp_${tag}_serial.log" 2>&1 || true )
python - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
rows = []
for path in glob.glob(f"gpurun_out/prof_supp_{tag}/**/*kernel_stats.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(path)) if "supp_" in r["Name"]]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:7.3f} ms")
PY
rm -rf "gpurun_out/prof_supp_${tag}"
