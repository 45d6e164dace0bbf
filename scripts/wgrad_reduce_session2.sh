#!/bin/bash
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/pf
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_hp.py tests/test_gpu_kernels.py -k "wgrad" -x -q > $out/tests_reduce.log 2>&1
echo "tests rc=$? $(tail -n 1 $out/tests_reduce.log)"
: > $out/micro_reduce.txt
for shape in "256 256 12 40" "512 512 6 20" "512 256 6 20" "256 128 24 80"; do
  for tm in 0 128; do
    echo -n "t_min=$tm " >> $out/micro_reduce.txt
    FP_WGRAD_REDUCE_T_MIN=$tm timeout 120 python scripts/wgrad_one.py $shape 12 40 2>/dev/null >> $out/micro_reduce.txt
  done
done
cat $out/micro_reduce.txt
: > $out/step_reduce.txt
for round in 1 2 3; do
  for cfg in "0:256" "128:256" "512:256"; do
    tm=${cfg%%:*}; tgt=${cfg##*:}
    echo -n "t_min=$tm target=$tgt " >> $out/step_reduce.txt
    FP_WGRAD_REDUCE_T_MIN=$tm FP_WGRAD_TARGET_WGS=$tgt timeout 300 python bench.py --leg train-only --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['img_per_s'], d['ms_per_step'], d['final_loss'])" >> $out/step_reduce.txt 2>&1
  done
done
cat $out/step_reduce.txt
