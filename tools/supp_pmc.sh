#!/bin/bash
# SQ counters of the suppressor's kernels run one after the other (no overlap): tools/supp_pmc.sh <tag> [lib.so]
set -e
cd "$(dirname "$0")/.."
tag="$1"; lib="${2:-audio-forge_amd/libaudioforge_mi.so}"
export AF_LIB_PATH="$PWD/$lib"
pass=0
for counters in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" \
                "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"; do
  pass=$((pass + 1))
  rm -rf "gpurun_out/pmc_${tag}_${pass}"
  ( cd /tmp && export TMPDIR=/tmp && AF_SERIAL_STREAMS=1 AF_SUPP_RAMP=0 AF_SUPP_WINDOW_FRAMES=50 AF_DIAG_SKIP_CHAIN=1 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_${tag}_${pass}" -- python "$OLDPWD/bench.py" --steps 1 --warmup 0 --seconds 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/pmc_${tag}_${pass}.log" 2>&1 || true )
  python tools/pmc_summary.py "gpurun_out/pmc_${tag}_${pass}" "gpurun_out/pmc_${tag}_${pass}.json" supp_ > /dev/null
  rm -rf "gpurun_out/pmc_${tag}_${pass}"
done
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
merged = {}
for p in (1, 2):
    for k, v in json.load(open(f"gpurun_out/pmc_{tag}_{p}.json")).items():
        merged.setdefault(k, {}).update({c: x["mean_per_dispatch"] for c, x in v.items()})
json.dump(merged, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1, sort_keys=True)
for k, v in sorted(merged.items()):
    wc = v.get("SQ_WAVE_CYCLES", 1.0)
    print(f"{tag} {k[:34]:34s} wave_cyc {wc:.3e} wait {v.get('SQ_WAIT_ANY',0)/wc:.2f} active {v.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} valu {v.get('SQ_ACTIVE_INST_VALU',0)/wc:.2f} "
          f"lds_act {v.get('SQ_ACTIVE_INST_LDS',0)/wc:.3f} conflict {v.get('SQ_LDS_BANK_CONFLICT',0)/wc:.3f} vmem_act {v.get('SQ_ACTIVE_INST_VMEM',0)/wc:.3f} "
          f"insts valu {v.get('SQ_INSTS_VALU',0):.3e} lds {v.get('SQ_INSTS_LDS',0):.3e} vmrd {v.get('SQ_INSTS_VMEM_RD',0):.3e} vmwr {v.get('SQ_INSTS_VMEM_WR',0):.3e} salu {v.get('SQ_INSTS_SALU',0):.3e}")
PY
