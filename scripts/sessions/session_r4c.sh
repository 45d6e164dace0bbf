#!/bin/bash
# GPU box, round 4 session C: whole GPU suite on the current build; step A/B against the previous build (before the branch-free ELU /
# 32-bit epilogue offsets / phase-kernel buffer loads) and against the v_fma_mix_f32-residual variant of the fp16-pair split
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r4c; mkdir -p $out; : > $out/summary.txt
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_all.log 2>&1; echo "pytest all rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
scripts/ab_lib_step.sh kitti rounds=2 prev default mix2 > $out/ab_step.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
scripts/ab_lib_step.sh matterport rounds=1 prev default > $out/ab_step_mp.txt 2>&1
timeout 300 python scripts/tile_bench.py default 30 > $out/tile_bench.txt 2>&1
echo "done t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
cat $out/summary.txt; tail -25 $out/pytest_all.log; cat $out/ab_step.txt $out/ab_step_mp.txt; cat $out/tile_bench.txt
