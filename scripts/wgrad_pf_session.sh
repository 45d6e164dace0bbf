#!/bin/bash
# GPU box: the weight gradient's register prefetch ring (FP_WGRAD_PF = ring depth, 0 = third-generation kernel): kernel tests per depth,
# per-shape timings, training-step A/B.  Results under gpurun_out/pf/.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/pf
mkdir -p $out
DEPTHS="${DEPTHS:-2 3}"
for pf in $DEPTHS; do
  FP_WGRAD_PF=$pf timeout 600 python -m pytest tests/test_gpu_hp.py -k "wgrad3x3" -x -q > $out/tests_pf$pf.log 2>&1
  echo "tests pf=$pf rc=$? $(tail -1 $out/tests_pf$pf.log)"
done
: > $out/micro.txt
for shape in "64 64 96 320" "64 64 48 160" "128 128 24 80" "256 256 12 40" "512 512 6 20" "128 64 48 160" "32 32 192 640"; do
  for pf in 0 $DEPTHS; do
    echo -n "pf=$pf " >> $out/micro.txt
    FP_WGRAD_PF=$pf timeout 120 python scripts/wgrad_one.py $shape 12 30 >> $out/micro.txt 2>&1
  done
done
cat $out/micro.txt
: > $out/step.txt
for round in 1 2; do
  for pf in 0 $DEPTHS; do
    echo -n "pf=$pf " >> $out/step.txt
    FP_WGRAD_PF=$pf timeout 300 python bench.py --leg train-only --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['img_per_s'], d['ms_per_step'], d['final_loss'])" >> $out/step.txt 2>&1
  done
done
cat $out/step.txt
