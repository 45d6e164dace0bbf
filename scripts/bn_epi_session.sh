#!/bin/bash
# GPU box: BatchNorm statistics out of the tile convolution's epilogue (FP_BN_EPI=1, the default) against the separate statistics pass:
# kernel test, network-level parity, training-step A/B.  Results under gpurun_out/bnepi/.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/bnepi
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_hp.py -k "batchnorm_partials or conv3x3_hp_kernel" -x -q > $out/tests_kernel.log 2>&1
echo "kernel tests rc=$? $(tail -n 1 $out/tests_kernel.log)"
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_switches.py -x -q > $out/tests_net.log 2>&1
echo "network tests rc=$? $(tail -n 1 $out/tests_net.log)"
: > $out/step.txt
for round in 1 2 3; do
  for e in 0 1; do
    echo -n "FP_BN_EPI=$e " >> $out/step.txt
    FP_BN_EPI=$e timeout 300 python bench.py --leg train-only --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['img_per_s'], d['ms_per_step'], d['final_loss'])" >> $out/step.txt 2>&1
  done
done
cat $out/step.txt
