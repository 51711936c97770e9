mkdir -p gpurun_out
for m in 1 2; do AF_ROLES=$m timeout -k 10 300 python -m pytest tests/test_gpu_roles.py -q 2>&1 | tail -2; done
for m in 0 1 2; do AF_ROLES=$m python bench.py --chain dynamics --seconds 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/dynr_$m.json 2>gpurun_out/dynr_$m.err; done
python - <<PY
import json
for n in ("0","1","2"):
    try:
        l=json.loads([x for x in open(f"gpurun_out/dynr_{n}.json") if x.startswith("{")][-1])
        print("AF_ROLES="+n, round(l["ms_per_step"],1), "ms per 2 s step", l["checks"]["output_rms"])
    except Exception as e: print(n, "failed", e)
PY
