#!/bin/bash
# GPU box, round 4 session E: remaining planner knobs (no rebuild) + the natural-statistics parity case over 8 seeds x 2 operand formats
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r4e; mkdir -p $out; : > $out/summary.txt
t0=$(date +%s)
scripts/ab_lib_step.sh kitti rounds=2 default default@FP_BN_ROWS_PER_THREAD=8 default@FP_BN_ROWS_PER_THREAD=16 default@FP_TILE_BN32_BELOW=250 default@FP_TILE_BN32_BELOW=400 default@FP_TILE_SK1_FROM=128 mix2 > $out/ab_step.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
timeout 1200 python scripts/natural_parity_seeds.py 8 4 > $out/natural_seeds.log 2>&1
echo "seeds done rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
cat $out/summary.txt $out/ab_step.txt; tail -30 $out/natural_seeds.log
