#!/bin/bash
# Round-3 profile artefacts (written to gpurun_out/, copied to profiles/ afterwards).  Two GPU calls of ~8 minutes each:
#   tools/r03_profiles.sh benches    bench lines (default full chain with the CPU baseline; auto-makeup; dynamics 4096 / 256;
#                                    de-esser 4096 / 256; full chain 256) + the counter passes of the default step
#   tools/r03_profiles.sh shapes     counter passes + kernel stats of the configs[1] shape and of the de-esser shape
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
summary() {
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03_*_bench.json")):
    try:
        l = json.loads([x for x in open(f) if x.startswith("{")][-1])
        print(f.split("/")[-1], round(l["ms_per_step"], 1), "ms", round(l["x_realtime"]), "x RT", l["config"]["kernel"][:40], "frac", round(l["roofline"]["frac"], 4), "traffic", l["roofline"].get("traffic"))
    except Exception as e:
        print(f, "failed", e)
PY
}
if [ "$1" != "shapes" ]; then
  python bench.py > gpurun_out/r03_full_chain_bench.json 2> gpurun_out/r03_full_chain_bench.err
  python bench.py --auto-makeup --no-cpu-baseline > gpurun_out/r03_full_chain_automakeup_bench.json 2> gpurun_out/r03_am.err
  python bench.py --chain dynamics --no-cpu-baseline > gpurun_out/r03_dynamics_bench.json 2> gpurun_out/r03_dyn.err
  python bench.py --chain dynamics --streams 256 --no-cpu-baseline > gpurun_out/r03_dynamics_256_bench.json 2> gpurun_out/r03_dyn256.err
  python bench.py --chain dynamics --deesser --no-cpu-baseline > gpurun_out/r03_deesser_4096_bench.json 2> gpurun_out/r03_de.err
  python bench.py --chain dynamics --deesser --streams 256 --no-cpu-baseline > gpurun_out/r03_deesser_256_bench.json 2> gpurun_out/r03_de256.err
  python bench.py --streams 256 --no-cpu-baseline > gpurun_out/r03_full_chain_256_bench.json 2> gpurun_out/r03_f256.err
  summary
  ROUND=r03 bash tools/step_counters.sh > gpurun_out/r03_stepc_full.log 2>&1
  cp gpurun_out/stepc_kernel_stats.csv gpurun_out/r03_full_chain_kernel_stats.csv
fi
if [ "$1" != "benches" ]; then
  ROUND=r03 bash tools/step_counters.sh --chain dynamics --streams 256 > gpurun_out/r03_stepc_dyn256.log 2>&1
  cp gpurun_out/stepc_kernel_stats.csv gpurun_out/r03_dynamics_256_kernel_stats.csv
  ROUND=r03 bash tools/step_counters.sh --chain dynamics --deesser > gpurun_out/r03_stepc_deesser.log 2>&1
  cp gpurun_out/stepc_kernel_stats.csv gpurun_out/r03_deesser_4096_kernel_stats.csv
fi
ls gpurun_out/r03_*counters*.json
