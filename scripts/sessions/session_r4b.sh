#!/bin/bash
# GPU box, round 4 session B: whole GPU suite on the current build; step A/B of the two-instruction fp16-pair split and of the BatchNorm
# backward sums out of the data-gradient epilogue
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r4b; mkdir -p $out; : > $out/summary.txt
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest_all.log 2>&1; echo "pytest all rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
scripts/ab_lib_step.sh kitti rounds=2 nomix default default@FP_BN_BWD_EPI=0 > $out/ab_step.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
timeout 300 python scripts/tile_bench.py default 30 > $out/tile_bench.txt 2>&1
timeout 300 python bench.py --workload matterport --no-loader --no-exact-split > $out/bench_matterport.json 2> $out/bench_matterport.err; echo "bench mp rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
cat $out/summary.txt; tail -15 $out/pytest_all.log; cat $out/ab_step.txt; cat $out/tile_bench.txt
