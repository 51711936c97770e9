#!/bin/bash
cd /root/repo 2>/dev/null || cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/full_$name.json 2> gpurun_out/full_$name.err; python - $name <<'PY'
import json, sys
n = sys.argv[1]
l = json.loads([x for x in open(f"gpurun_out/full_{n}.json") if x.startswith("{")][-1])
print(f"{n}: {l['ms_per_step']:.1f} ms/step  supp {l['stage_ms']['suppressor_and_front_end']:.1f}  chain(sum) {l['stage_ms']['chain']:.1f}  rms {l['checks']['output_rms']:.9f}")
PY
}
run default AF_X=1
run detk AF_TP_DETECT_KERNEL=1
run win16 AF_SUPP_WINDOW_FRAMES=16
run win30 AF_SUPP_WINDOW_FRAMES=30
run win40 AF_SUPP_WINDOW_FRAMES=40
run detk30 AF_TP_DETECT_KERNEL=1 AF_SUPP_WINDOW_FRAMES=30
run default2 AF_X=1
