#!/bin/bash
# GPU box: amax slots in one vector round trip, issued before / reduced behind the first operand loads (tile, weight-gradient, phase, igemm kernels)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4o; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_hp.py tests/test_gpu_kernels.py -x -q > $O/pytest_kernels.log 2>&1
echo "pytest kernels rc=$? t=$(( $(date +%s)-t0 ))"
bash scripts/ab_lib_step.sh kitti rounds=3 default preamax3 > $O/ab_step.txt 2>&1
bash scripts/ab_lib_step.sh matterport rounds=1 default preamax3 > $O/ab_step_mp.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))"
timeout 300 python scripts/tile_bench.py default > $O/tile_bench.txt 2>&1
FP_LIB=$R/scripts/ubench/bin/lib_preamax3.so timeout 300 python scripts/tile_bench.py preamax3 >> $O/tile_bench.txt 2>&1
echo "tile bench done t=$(( $(date +%s)-t0 ))"
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_trainer.py -x -q > $O/pytest_net.log 2>&1
echo "pytest net rc=$? t=$(( $(date +%s)-t0 ))"
tail -3 $O/pytest_kernels.log; tail -3 $O/pytest_net.log; cat $O/ab_step.txt $O/ab_step_mp.txt; cat $O/tile_bench.txt
