#!/bin/bash
# GPU box: tile-form weight repack (+ bounded side-stream launch, forward / data-gradient tables), paired weight-gradient forks, scatter-form head weight gradient
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4i; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "pack or head" > $O/pytest_kernels.log 2>&1
echo "pytest pack/head rc=$? t=$(( $(date +%s)-t0 ))"
timeout 900 python -m pytest tests/test_gpu_hp.py tests/test_gpu_network.py -x -q > $O/pytest_net.log 2>&1
echo "pytest hp+net rc=$? t=$(( $(date +%s)-t0 ))"
timeout 300 python scripts/hbm_microbench.py > $O/hbm_kernels.txt 2>&1
echo "hbm done t=$(( $(date +%s)-t0 ))"
bash scripts/ab_lib_step.sh kitti rounds=2 default default@FP_PACK_TILED=0,FP_WGRAD_PAIR_FORK=0 default@FP_PACK_SIDE_WGS=0 default@FP_PACK_SIDE_WGS=128 default@FP_PACK_SIDE_WGS=512 default@FP_WGRAD_PAIR_FORK=0 > $O/ab_step.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats -d /tmp/pc -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/pc -name "*.db" | head -1) -2 trace > $O/trace_step.txt 2>&1
echo "trace done t=$(( $(date +%s)-t0 ))"
cd $R
timeout 600 python -m pytest "tests/test_gpu_parity_fullsize.py::test_train_step_fp64_anchored[12-192-640]" -x -q -s > $O/pytest_decomp.log 2>&1
echo "pytest parity rc=$? t=$(( $(date +%s)-t0 ))"
tail -3 $O/pytest_kernels.log; tail -3 $O/pytest_net.log; cat $O/ab_step.txt; grep -i "head_wgrad\|loss\|maxpool" $O/hbm_kernels.txt; head -3 $O/trace_step.txt; tail -4 $O/pytest_decomp.log
