#!/bin/bash
# GPU box: training-step A/B of the weight gradient's prefetch ring depths (alternating runs).  Results in gpurun_out/pf/step.txt
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/pf
mkdir -p $out
: > $out/step.txt
for round in 1 2; do
  for cfg in ${CFGS:-"0:256" "2:256" "3:256" "2:512"}; do
    pf=${cfg%%:*}; tgt=${cfg##*:}
    echo -n "pf=$pf target=$tgt " >> $out/step.txt
    FP_WGRAD_PF=$pf FP_WGRAD_TARGET_WGS=$tgt timeout 300 python bench.py --leg train-only --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['img_per_s'], d['ms_per_step'], d['final_loss'])" >> $out/step.txt 2>&1
  done
done
cat $out/step.txt
