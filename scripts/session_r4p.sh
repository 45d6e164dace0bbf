#!/bin/bash
# GPU box: BatchNorm statistics of the split-K levels out of the reduce launch (tile kernel and fp16-pair implicit GEMM)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4p; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_hp.py tests/test_gpu_kernels.py -x -q > $O/pytest_kernels.log 2>&1
echo "pytest kernels rc=$? t=$(( $(date +%s)-t0 ))"
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_trainer.py tests/test_gpu_segmentation.py tests/test_gpu_switches.py -x -q > $O/pytest_net.log 2>&1
echo "pytest net rc=$? t=$(( $(date +%s)-t0 ))"
bash scripts/ab_lib_step.sh kitti rounds=3 default prestats > $O/ab_step.txt 2>&1
bash scripts/ab_lib_step.sh matterport rounds=1 default prestats > $O/ab_step_mp.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))"
tail -3 $O/pytest_kernels.log; tail -3 $O/pytest_net.log; cat $O/ab_step.txt $O/ab_step_mp.txt
