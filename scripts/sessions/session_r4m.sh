#!/bin/bash
# GPU box: heads' weight gradient fused into their data gradient; stem forward with the K-offset table
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4m; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "head or stem" > $O/pytest_kernels.log 2>&1
echo "pytest kernels rc=$? t=$(( $(date +%s)-t0 ))"
timeout 300 python scripts/hbm_microbench.py > $O/hbm_kernels.txt 2>&1
bash scripts/ab_lib_step.sh kitti rounds=2 default default@FP_HEAD_FUSED=0 > $O/ab_step.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps; FP_SERIAL=1 FP_PLAN=0 rocprofv3 --kernel-trace --stats -d /tmp/ps -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
python $R/scripts/rocprof_summary.py $(find /tmp/ps -name "*.db" | head -1) $O/kernel_stats_serial.txt "serial" > /dev/null 2>&1
echo "serial stats done t=$(( $(date +%s)-t0 ))"
cd $R
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_trainer.py tests/test_gpu_segmentation.py -x -q > $O/pytest_net.log 2>&1
echo "pytest net rc=$? t=$(( $(date +%s)-t0 ))"
tail -3 $O/pytest_kernels.log; tail -3 $O/pytest_net.log; cat $O/ab_step.txt; grep -i "head_" $O/hbm_kernels.txt; grep -i "stem\|head_" $O/kernel_stats_serial.txt
