#!/bin/bash
# copies what scripts/profile_r3.sh left under gpurun_out/prof_r3/ into profiles/r3/ (the tracked, judged copies)
set -e
cd "$(dirname "$0")/.."
P=gpurun_out/prof_r3; D=profiles/r3
for w in cornell default grid; do
  cp $P/bench_under_rocprof_$w.json $D/bench_${w}_under_rocprof.json
  cp $P/${w}_default_cmd_kernel_stats.csv $P/${w}_default_cmd_steady.csv $D/
  cp $P/lane_profile_$w.log $D/
  cp $P/pmc_$w.txt $D/pmc_${w}_wavestack.txt
done
cp $P/bench_default_cmd.json $P/pmc_traffic.json $D/
cp gpurun_out/profile_r3.txt $D/profile_r3.txt
python3 scripts/code_object.py > $D/code_object.txt 2>&1
