#!/bin/bash
# GPU box: the driver's command once more on the final tree (test_g2 now prints where it leaves the oracle when it fails)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5k; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? t=$(( $(date +%s)-t0 ))" > $O/summary.txt
grep -B2 -A16 "g2 diagnose" $O/pytest_gpu.log | head -60
tail -5 $O/pytest_gpu.log
cp -r gpurun_out/parity $O/ 2>/dev/null
cat $O/summary.txt
