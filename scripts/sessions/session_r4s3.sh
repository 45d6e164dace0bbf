#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_hp.py tests/test_gpu_network.py -x -q -k "stem or full or train" > $O/pytest_stem3.log 2>&1
echo "pytest stem rc=$?"
bash scripts/ab_lib_step.sh kitti rounds=2 default default@FP_HP_STEM=0 default@FP_STEM_WGRAD_HP_WGS=768 > $O/ab_step3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps; FP_SERIAL=1 FP_PLAN=0 rocprofv3 --kernel-trace --stats -d /tmp/ps -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
python $R/scripts/rocprof_summary.py $(find /tmp/ps -name "*.db" | head -1) $O/kernel_stats_serial3.txt "serial" > /dev/null 2>&1
cd $R
tail -2 $O/pytest_stem3.log; cat $O/ab_step3.txt; grep -i "stem" $O/kernel_stats_serial3.txt
