#!/bin/bash
# usage: run_variants.sh name...
for v in "$@"; do
  export MIPT_LIB=$PWD/cudapathtracer_amd/libmipt_$v.so
  [ "$v" = base ] && export MIPT_LIB=$PWD/cudapathtracer_amd/libmipt.so
  line="$v"
  for w in cornell default grid; do
    timeout -k 10 200 python bench.py --workload $w --cpu-seconds 0 --trace-loop-frames 0 --closest-hit-only 0 --min-seconds 2 > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "$v $w failed"; tail -2 gpurun_out/bv.err; exit 1; }
    line="$line $(python -c "import json; d=json.loads(open('gpurun_out/bv.json').read().strip().splitlines()[-1]); print(d['value'])")"
  done
  echo "$line"
done
