#!/bin/bash
# Same-box sweep of the suppressor pipeline's knobs on the bench step: tools/full_sweep.sh [bench args]
run() { echo -n "$1: "; env $2 python bench.py --no-cpu-baseline --steps 15 --warmup 3 "${@:3}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1), round(d['roofline']['avg_kernel_ms'],2), {k: round(v,1) for k,v in d['stage_ms'].items()})"; }
run base "AF_X=0" "$@"
run synth_split0 "AF_SYNTH_SPLIT=0" "$@"
for w in 6 8 12 16 24 30 40; do run win$w "AF_SUPP_WINDOW_FRAMES=$w" "$@"; done
run noramp "AF_SUPP_RAMP=0" "$@"
run base2 "AF_X=0" "$@"
