#!/bin/bash
# GPU box, final build of round 4: the whole GPU suite, smoke, the N = 2 shared-GPU bench line, FP_HP=0 network tests
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r4t; mkdir -p $out; : > $out/summary.txt
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_all.log 2>&1; echo "pytest all rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
FP_HP=0 timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_trainer.py -q > $out/pytest_exact.log 2>&1; echo "pytest exact rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
timeout 600 python bench.py --gpus 2 --steps 6 --warmup 3 --no-cpu-baseline > $out/bench_gpus2_shared.json 2> $out/bench_gpus2.err; echo "bench gpus2 rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
cat $out/summary.txt; tail -6 $out/pytest_all.log; tail -4 $out/pytest_exact.log; tail -2 $out/smoke.log; cat gpurun_out/parity/parity_ratios_hp.md
