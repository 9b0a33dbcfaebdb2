#!/bin/bash
# PMC counter passes for the bench workload, one rocprofv3 run per pass (counters only, no trace domains).
# usage: scripts/pmc_passes.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/<pass>/...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {   # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --min-seconds 0 "${BENCH_ARGS[@]}" > $OUT/$name.log 2>&1
  local rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES || exit 1
run sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32 || exit 1
run tcc1 FETCH_SIZE TCC_HIT_sum || exit 1
run tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum || exit 1
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE || exit 1
echo ALL_PASSES_DONE
