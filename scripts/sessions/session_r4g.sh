#!/bin/bash
# GPU box, round 4 session G: the whole GPU suite on the final build (default operand format), the full-size parity cases with exactly split
# operands (FP_HP=0), the N = 2 bench line on the shared GPU (launch path + fields of the line, not a performance number)
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r4g; mkdir -p $out; : > $out/summary.txt
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_all.log 2>&1; echo "pytest all rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
FP_HP=0 timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py tests/test_gpu_network.py -q > $out/pytest_exact.log 2>&1; echo "pytest exact rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
timeout 600 python bench.py --gpus 2 --steps 6 --warmup 3 --no-cpu-baseline > $out/bench_gpus2_shared.json 2> $out/bench_gpus2.err; echo "bench gpus2 rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
cat $out/summary.txt; tail -6 $out/pytest_all.log; tail -4 $out/pytest_exact.log; tail -c 1500 $out/bench_gpus2_shared.json; tail -2 $out/smoke.log; cat gpurun_out/parity/parity_ratios_*.md
