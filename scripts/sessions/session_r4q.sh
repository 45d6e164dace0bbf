#!/bin/bash
# GPU box: BatchNorm backward sums of the split-K data gradients out of the reduce launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_hp.py -x -q > $O/pytest_hp.log 2>&1
echo "pytest hp rc=$? t=$(( $(date +%s)-t0 ))"
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_trainer.py tests/test_gpu_dp.py -x -q > $O/pytest_net.log 2>&1
echo "pytest net rc=$? t=$(( $(date +%s)-t0 ))"
bash scripts/ab_lib_step.sh kitti rounds=3 default prestats default@FP_BN_BWD_EPI=0 > $O/ab_step.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats -d /tmp/pc -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/pc -name "*.db" | head -1) -2 trace > $O/trace_step.txt 2>&1
cd $R
tail -3 $O/pytest_hp.log; tail -3 $O/pytest_net.log; cat $O/ab_step.txt; head -3 $O/trace_step.txt
