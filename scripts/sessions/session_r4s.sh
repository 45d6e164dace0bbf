#!/bin/bash
# GPU box: the stem with fp16-pair operands (forward + weight gradient)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_hp.py -x -q -k "stem or pack" > $O/pytest_stem.log 2>&1
echo "pytest stem rc=$? t=$(( $(date +%s)-t0 ))"
timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_trainer.py -x -q > $O/pytest_net.log 2>&1
echo "pytest net rc=$? t=$(( $(date +%s)-t0 ))"
bash scripts/ab_lib_step.sh kitti rounds=3 default default@FP_HP_STEM=0 > $O/ab_step.txt 2>&1
bash scripts/ab_lib_step.sh matterport rounds=1 default default@FP_HP_STEM=0 > $O/ab_step_mp.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats -d /tmp/pc -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
python $R/scripts/timeline.py $(find /tmp/pc -name "*.db" | head -1) -2 trace > $O/trace_step.txt 2>&1
cd $R
tail -5 $O/pytest_stem.log; tail -3 $O/pytest_net.log; cat $O/ab_step.txt $O/ab_step_mp.txt; grep -i "stem" $O/trace_step.txt | head
