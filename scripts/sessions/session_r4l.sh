#!/bin/bash
# GPU box: the whole GPU suite on the lazily repacked fp32 layouts (+ FP_HP=0 network tests), A/B of the lazy repack, ordered trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4l; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1
echo "pytest all rc=$? t=$(( $(date +%s)-t0 ))"
FP_HP=0 timeout 600 python -m pytest tests/test_gpu_network.py tests/test_gpu_trainer.py -x -q > $O/pytest_exact.log 2>&1
echo "pytest exact rc=$? t=$(( $(date +%s)-t0 ))"
bash scripts/ab_lib_step.sh kitti rounds=2 default default@FP_PACK_LAZY32=0 default@FP_PACK_LAZY32=0,FP_PACK_TILED=0,FP_WGRAD_PAIR_FORK=0 > $O/ab_step.txt 2>&1
bash scripts/ab_lib_step.sh matterport rounds=1 default default@FP_PACK_LAZY32=0,FP_PACK_TILED=0,FP_WGRAD_PAIR_FORK=0 > $O/ab_step_mp.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats -d /tmp/pc -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/pc -name "*.db" | head -1) -2 trace > $O/trace_step.txt 2>&1
echo "trace done t=$(( $(date +%s)-t0 ))"
cd $R
tail -4 $O/pytest_all.log; tail -3 $O/pytest_exact.log; cat $O/ab_step.txt $O/ab_step_mp.txt; head -3 $O/trace_step.txt
