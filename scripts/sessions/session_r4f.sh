#!/bin/bash
# GPU box, round 4 session F: the wave-specialised weight gradient (FP_WGRAD_WS=1) -- kernel tests, per-shape times, step A/B
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r4f; mkdir -p $out; : > $out/summary.txt
t0=$(date +%s)
FP_WGRAD_WS=1 timeout 600 python -m pytest tests/test_gpu_hp.py tests/test_gpu_kernels.py -x -q -k "wgrad" > $out/pytest_ws.log 2>&1; echo "pytest ws rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
: > $out/wgrad_one.txt
for shape in "64 64 96 320" "64 64 48 160" "128 128 24 80" "256 256 12 40" "512 512 6 20" "128 64 48 160" "32 32 192 640"; do
  for ws in 0 1; do echo -n "ws=$ws " >> $out/wgrad_one.txt; FP_WGRAD_WS=$ws timeout 120 python scripts/wgrad_one.py $shape 12 20 2>&1 | tail -1 >> $out/wgrad_one.txt; done
done
echo "bench done t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
scripts/ab_lib_step.sh kitti rounds=2 default default@FP_WGRAD_WS=1 default@FP_WGRAD_WS=1,FP_WGRAD_TARGET_WGS=384 > $out/ab_step.txt 2>&1
scripts/ab_lib_step.sh matterport rounds=1 default default@FP_WGRAD_WS=1 > $out/ab_step_mp.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
FP_WGRAD_WS=1 timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py tests/test_gpu_network.py -x -q > $out/pytest_ws_net.log 2>&1; echo "pytest ws net rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
cat $out/summary.txt; tail -5 $out/pytest_ws.log; cat $out/wgrad_one.txt $out/ab_step.txt $out/ab_step_mp.txt; tail -5 $out/pytest_ws_net.log
