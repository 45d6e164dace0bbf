#!/bin/bash
# GPU box: the whole-suite failure of test_g2 (sessions G): does the list of files in front of it reproduce it when it is the FIRST process on a fresh box?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5i; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_data_path.py tests/test_gpu_dp.py tests/test_gpu_golden_blocks.py tests/test_gpu_hp.py tests/test_gpu_kernels.py tests/test_gpu_network.py -q -p no:cacheprovider -s > $O/first.log 2>&1; echo "first rc=$? t=$(( $(date +%s)-t0 ))" > $O/summary.txt
grep -A12 "g2 diagnose" $O/first.log | head -40
grep -E "passed|failed" $O/first.log | tail -1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -s -k "not parity_fullsize and not switches and not trainer and not segmentation" > $O/second.log 2>&1; echo "second(-m gpu) rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
grep -A12 "g2 diagnose" $O/second.log | head -40
grep -E "passed|failed" $O/second.log | tail -1
cat $O/summary.txt
