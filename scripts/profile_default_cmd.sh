#!/bin/bash
# rocprofv3 kernel trace of bench.py exactly as the driver runs it (default steps / warm-up), per workload, with the
# steady-state summary of scripts/kt_steady.py beside rocprofv3's own --stats table.
# usage: scripts/profile_default_cmd.sh <round> [workloads...]     (on the GPU box; results in gpurun_out/kt_<round>/)
set -u
ROUND=$1; shift
WL=("$@"); [ ${#WL[@]} -eq 0 ] && WL=(cornell default grid)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kt_$ROUND
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export PYTHONPATH=$R
for w in "${WL[@]}"; do
  echo "== $w"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw_$w -o kt -- python3 $R/bench.py --workload $w > $OUT/bench_under_rocprof_$w.json 2> $OUT/bench_under_rocprof_$w.err || { echo "kernel trace failed for $w"; exit 1; }
  cp $(find $OUT/raw_$w -name '*kernel_stats.csv' | head -1) $OUT/${w}_default_cmd_kernel_stats.csv
  python3 $R/scripts/kt_steady.py $(find $OUT/raw_$w -name '*kernel_trace.csv' | head -1) 4 > $OUT/${w}_default_cmd_steady.csv || exit 1
  cat $OUT/${w}_default_cmd_steady.csv
  python3 -c "
import json; d=json.loads(open('$OUT/bench_under_rocprof_$w.json').read().strip().splitlines()[-1]); r=d['roofline']
print(' bench under rocprofv3:', d['value'], 'Mrays/s, events', r['avg_launch_ms'], 'ms/launch, wall', r.get('wall_ms_per_launch'), 'ms/launch, launches', r['launches'])"
  rm -rf $OUT/raw_$w
done
echo KT_DONE
