#!/bin/bash
# PMC passes (scripts/pmc_passes.sh) of one workload for the shipped library and one variant library, side by side.
# usage: scripts/pmc_ab.sh <workload> <variant.so under cudapathtracer_amd/>   -> gpurun_out/pmc_ab.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=$1; V=$2
bash $R/scripts/pmc_passes.sh ab_base --workload $W --trace-loop-frames 0 2>&1 | tail -2
MIPT_LIB=$R/cudapathtracer_amd/$V bash $R/scripts/pmc_passes.sh ab_var --workload $W --trace-loop-frames 0 2>&1 | tail -2
python3 $R/scripts/pmc_to_json.py $R/gpurun_out/pmc_ab.json base=$R/gpurun_out/pmc_ab_base variant=$R/gpurun_out/pmc_ab_var > /dev/null
python3 - <<PY
import json
d = json.load(open("$R/gpurun_out/pmc_ab.json"))
for k in ("base", "variant"):
    e = d[k]
    print(k, {x: e.get(x) for x in ("valu_busy", "lane_utilisation", "wait_frac", "l2_hit", "valu_wave_instructions_per_launch", "executed_lane_ops_per_launch", "hbm_gb_per_launch")})
PY
