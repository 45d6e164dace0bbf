#!/bin/bash
# GPU box: pipelined tile repack, placement of the data-gradient repack, head weight gradient with the interior fast path, wide stem reduce
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4j; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "pack or head or wgrad or stem" > $O/pytest_kernels.log 2>&1
echo "pytest kernels rc=$? t=$(( $(date +%s)-t0 ))"
timeout 300 python scripts/hbm_microbench.py > $O/hbm_kernels.txt 2>&1
echo "hbm done t=$(( $(date +%s)-t0 ))"
bash scripts/ab_lib_step.sh kitti rounds=2 default default@FP_PACK_TILED=0,FP_WGRAD_PAIR_FORK=0 default@FP_PACK_SIDE_WGS=512 default@FP_PACK_SIDE_WGS=1024 default@FP_PACK_DGRAD_LATE=1 default@FP_PACK_DGRAD_LATE=1,FP_PACK_SIDE_WGS=512 > $O/ab_step.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats -d /tmp/pc -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/pc -name "*.db" | head -1) -2 trace > $O/trace_step.txt 2>&1
echo "trace done t=$(( $(date +%s)-t0 ))"
cd $R
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_trainer.py -x -q > $O/pytest_net.log 2>&1
echo "pytest net rc=$? t=$(( $(date +%s)-t0 ))"
tail -3 $O/pytest_kernels.log; tail -3 $O/pytest_net.log; cat $O/ab_step.txt; grep -i "head_wgrad" $O/hbm_kernels.txt; head -3 $O/trace_step.txt
