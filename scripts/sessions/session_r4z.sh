#!/bin/bash
# GPU box, final tree of round 4: the whole GPU suite + smoke + one default bench line
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r4z; mkdir -p $out; : > $out/summary.txt
t0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q > $out/pytest_all.log 2>&1; echo "pytest all rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
cat $out/summary.txt; tail -3 $out/pytest_all.log; tail -1 $out/smoke.log; cat $out/bench_default.json
