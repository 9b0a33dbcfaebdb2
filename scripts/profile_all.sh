#!/bin/bash
# Regenerates the evidence under gpurun_out/prof_<round>/ for the three bench workloads: bench JSON line,
# rocprofv3 kernel-trace stats of the same command, PMC passes (scripts/pmc_passes.sh) and their summaries.
# usage: scripts/profile_all.sh <round> [workloads...]      (run on the GPU box; copy what is judged into profiles/<round>/)
set -u
ROUND=$1; shift
WL=("$@"); [ ${#WL[@]} -eq 0 ] && WL=(cornell default grid)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$ROUND
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export PYTHONPATH=$R
for w in "${WL[@]}"; do
  steps=8; [ $w = grid ] && steps=4
  echo "== $w: kernel trace"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$w -o kt -- python3 $R/bench.py --workload $w --steps $steps --warmup 2 --cpu-seconds 0 > $OUT/kt_$w.log 2>&1 || { echo "kernel trace failed for $w"; exit 1; }
  cp $(find $OUT/kt_$w -name '*kernel_stats.csv' | head -1) $OUT/${w}_kernel_stats.csv
  echo "== $w: pmc"
  bash $R/scripts/pmc_passes.sh ${ROUND}_$w --workload $w || { echo "pmc failed for $w"; exit 1; }
  python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_${ROUND}_$w > $OUT/pmc_${w}.txt
done
args=(); for w in "${WL[@]}"; do args+=("$w=$R/gpurun_out/pmc_${ROUND}_$w"); done
python3 $R/scripts/pmc_to_json.py $OUT/pmc_traffic.json "${args[@]}" > /dev/null
mkdir -p $R/profiles/$ROUND && cp $OUT/pmc_traffic.json $R/profiles/$ROUND/pmc_traffic.json   # bench.py reads it from there
for w in "${WL[@]}"; do
  steps=8; [ $w = grid ] && steps=4
  echo "== $w: bench"
  timeout -k 10 400 python3 $R/bench.py --workload $w --steps $steps --warmup 2 > $OUT/bench_$w.json 2> $OUT/bench_$w.err || { echo "bench failed for $w"; exit 1; }
  tail -c 2500 $OUT/bench_$w.json
done
# the same without the frame pipeline: one launch at a time, so HIP-event windows and rocprofv3 durations are those of
# the kernel alone and must agree
for w in "${WL[@]}"; do
  steps=8; [ $w = grid ] && steps=4
  echo "== $w: serial launches (--pipeline 0)"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts_$w -o kt -- python3 $R/bench.py --workload $w --steps $steps --warmup 2 --cpu-seconds 0 --pipeline 0 > $OUT/kts_$w.log 2>&1 || { echo "serial kernel trace failed for $w"; exit 1; }
  cp $(find $OUT/kts_$w -name '*kernel_stats.csv' | head -1) $OUT/${w}_serial_kernel_stats.csv
  timeout -k 10 400 python3 $R/bench.py --workload $w --steps $steps --warmup 2 --pipeline 0 > $OUT/bench_${w}_serial.json 2> $OUT/bench_${w}_serial.err || { echo "serial bench failed for $w"; exit 1; }
done
if [ -f $R/cudapathtracer_amd/libmipt_prof.so ]; then
  for w in "${WL[@]}"; do
    echo "== $w: lane profile (instrumented build)"
    MIPT_LIB=$R/cudapathtracer_amd/libmipt_prof.so timeout -k 10 200 python3 $R/scripts/lane_profile.py $w > $OUT/lane_profile_$w.log 2>&1 || { echo "lane profile failed for $w"; exit 1; }
    cat $OUT/lane_profile_$w.log
  done
fi
echo PROFILE_ALL_DONE
