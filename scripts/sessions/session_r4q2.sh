#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_hp.py tests/test_gpu_network.py -x -q > $O/pytest_hp2.log 2>&1
echo "pytest rc=$?"
bash scripts/ab_lib_step.sh kitti rounds=3 default prestats > $O/ab_step2.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats -d /tmp/pc -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/pc -name "*.db" | head -1) -2 trace > $O/trace_step2.txt 2>&1
cd $R
tail -3 $O/pytest_hp2.log; cat $O/ab_step2.txt; grep -i "splitk\|bn_bwd_reduce\|bn_stats_kernel" $O/trace_step2.txt | head
