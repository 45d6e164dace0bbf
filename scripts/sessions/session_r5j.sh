#!/bin/bash
# GPU box: the whole-suite failure of test_g2 again -- the exact command of session G (output capture on, --durations) cut off behind test_gpu_network.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5j; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider -k "not parity_fullsize and not switches and not trainer and not segmentation" > $O/capture_on.log 2>&1; echo "capture_on rc=$? t=$(( $(date +%s)-t0 ))" > $O/summary.txt
grep -A14 "g2 diagnose" $O/capture_on.log | head -40
grep -E "passed|failed" $O/capture_on.log | tail -1
timeout 900 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider -k "not parity_fullsize and not switches and not trainer and not segmentation" > $O/capture_on_2.log 2>&1; echo "capture_on_again rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
grep -A14 "g2 diagnose" $O/capture_on_2.log | head -40
grep -E "passed|failed" $O/capture_on_2.log | tail -1
cat $O/summary.txt
