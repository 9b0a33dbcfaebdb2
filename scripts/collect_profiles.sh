#!/bin/bash
# Copies what scripts/profile_all.sh left under gpurun_out/prof_<round>/ into profiles/<round>/ (the judged, tracked copies).
R=${1:-r1}; P=gpurun_out/prof_$R; D=profiles/$R
set -e
cp $P/cornell_kernel_stats.csv $D/cornell_1080p_d8_kernel_stats.csv
cp $P/default_kernel_stats.csv $D/default_1080p_d12_kernel_stats.csv
cp $P/grid_kernel_stats.csv $D/grid10m_4k_d4_kernel_stats.csv
cp $P/pmc_cornell.txt $D/pmc_cornell_1080p_d8_wavestack.txt
cp $P/pmc_default.txt $D/pmc_default_1080p_d12_wavestack.txt
cp $P/pmc_grid.txt $D/pmc_grid10m_4k_d4_wavestack.txt
cp $P/pmc_traffic.json $D/
cp $P/bench_cornell.json $D/bench_cornell_wavestack.json
cp $P/bench_default.json $D/bench_default_wavestack.json
cp $P/bench_grid.json $D/bench_grid10m_wavestack.json
cp $P/lane_profile_*.log $D/
for w in cornell default grid; do cp $P/${w}_serial_kernel_stats.csv $D/${w}_serial_kernel_stats.csv; cp $P/bench_${w}_serial.json $D/bench_${w}_serial.json; done
for w in cornell default grid; do head -2 $P/${w}_kernel_stats.csv | tail -1 | cut -c1-110; python3 -c "
import json,sys; d=json.load(open('$P/bench_$w.json')); r=d['roofline']; print(' bench', d['value'], 'Mrays/s', r['avg_launch_ms'], 'ms/launch', 'B/ray', r['bytes_per_ray'], 'N', r['n_node_per_ray'], r['n_tri_per_ray'], 'achieved', r['achieved'], 'frac', r['frac'], 'traffic', r['traffic'], 'cpu', d['cpu_baseline']['value'])"; done
for w in cornell default grid; do head -2 $P/${w}_serial_kernel_stats.csv | tail -1 | cut -d, -f1-5 | cut -c30-; python3 -c "
import json; d=json.load(open('$P/bench_${w}_serial.json')); r=d['roofline']; print(' serial bench', d['value'], 'Mrays/s', r['avg_launch_ms'], 'ms/launch achieved', r['achieved'], 'frac', r['frac'])"; done
