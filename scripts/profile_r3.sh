#!/bin/bash
# Round-3 evidence (GPU box): for each bench workload the rocprofv3 kernel trace of bench.py's own measurement (default steps
# and warm-up), the PMC passes, then the driver's default command itself with the fresh PMC figures attached, and the lane
# profile of the instrumented build.  Results in gpurun_out/prof_r3/; copy what is judged into profiles/r3/.
# usage: scripts/profile_r2.sh [workloads...]
set -u
WL=("$@"); [ ${#WL[@]} -eq 0 ] && WL=(cornell default grid)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r3
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export PYTHONPATH=$R
for w in "${WL[@]}"; do
  echo "== $w: kernel trace of the bench measurement"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw_$w -o kt -- python3 $R/bench.py --workload $w --cpu-seconds 0 --trace-loop-frames 0 --closest-hit-only 0 --min-seconds 2 > $OUT/bench_under_rocprof_$w.json 2> $OUT/bench_under_rocprof_$w.err || { echo "kernel trace failed for $w"; tail -3 $OUT/bench_under_rocprof_$w.err; exit 1; }
  cp $(find $OUT/raw_$w -name '*kernel_stats.csv' | head -1) $OUT/${w}_default_cmd_kernel_stats.csv
  python3 $R/scripts/kt_steady.py $(find $OUT/raw_$w -name '*kernel_trace.csv' | head -1) 4 k_wavestack 32 > $OUT/${w}_default_cmd_steady.csv || exit 1
  cat $OUT/${w}_default_cmd_steady.csv
  python3 -c "
import json; d=json.loads(open('$OUT/bench_under_rocprof_$w.json').read().strip().splitlines()[-1])
print(' bench under rocprofv3:', d['value'], 'Mrays/s,', d['ms_per_step'], 'ms/step, kernel_window_ms (mean event window)', d['kernel_window_ms'], 'kernel_span_ms_per_step', d['kernel_span_ms_per_step'])"
  rm -rf $OUT/raw_$w
  echo "== $w: pmc"
  bash $R/scripts/pmc_passes.sh r3_$w --workload $w --trace-loop-frames 0 --closest-hit-only 0 || { echo "pmc failed for $w"; exit 1; }
  python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_r3_$w > $OUT/pmc_${w}.txt
done
args=(); for w in "${WL[@]}"; do args+=("$w=$R/gpurun_out/pmc_r3_$w"); done
python3 $R/scripts/pmc_to_json.py $OUT/pmc_traffic.json "${args[@]}" > /dev/null || exit 1
mkdir -p $R/profiles/r3 && cp $OUT/pmc_traffic.json $R/profiles/r3/pmc_traffic.json   # bench.py attaches it (same sources only)
for w in "${WL[@]}"; do rm -rf $R/gpurun_out/pmc_r3_$w/*/; done
echo "== the driver's default command"
timeout -k 10 600 python3 $R/bench.py > $OUT/bench_default_cmd.json 2> $OUT/bench_default_cmd.err || { echo "default bench failed"; tail -3 $OUT/bench_default_cmd.err; exit 1; }
python3 -c "
import json; d=json.loads(open('$OUT/bench_default_cmd.json').read().strip().splitlines()[-1])
for n, w in [('cornell', d)] + list(d['workloads'].items()):
    r = w['roofline']; print(n, w['value'], 'Mrays/s', w['ms_per_step'], 'ms/step; launch', r['avg_launch_ms'], 'ms; bound', r['bound'], 'frac', r['frac'], 'valu', r['valu']['frac'], 'hbm', r['hbm']['frac'], 'traffic', r['traffic'], 'traffic_frac', r['hbm']['traffic_frac'], 'cpu', w['cpu_baseline']['value'], 'trace_loop ms/frame', w['trace_loop']['ms_per_frame'])"
if [ -f $R/cudapathtracer_amd/libmipt_prof.so ]; then
  for w in "${WL[@]}"; do
    MIPT_LIB=$R/cudapathtracer_amd/libmipt_prof.so timeout -k 10 200 python3 $R/scripts/lane_profile.py $w > $OUT/lane_profile_$w.log 2>&1 || { echo "lane profile failed for $w"; exit 1; }
    cat $OUT/lane_profile_$w.log
  done
fi
echo PROFILE_R3_DONE
