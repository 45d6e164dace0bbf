#!/bin/bash
# GPU box: test_g2's failure in the driver's command (block2.pre_concat_conv.conv1 of the 4 x 6 level, fp32 flattened kernel): which mechanism?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5l; mkdir -p $O
cd $R
t0=$(date +%s)
for cfg in "FP_NO_SPLITK=1" "FP_PACK_LAZY32=0" "X=1"; do
  env $cfg timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_parity_fullsize.py > $O/run_$cfg.log 2>&1; echo "$cfg rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
  grep -A8 "g2 diagnose" $O/run_$cfg.log | head -12
  grep -E "passed|failed" $O/run_$cfg.log | tail -1
done
cat $O/summary.txt
