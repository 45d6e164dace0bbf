#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_hp.py -x -q -k "stem" > $O/pytest_stem2.log 2>&1
echo "pytest stem rc=$?"
bash scripts/ab_lib_step.sh kitti rounds=2 default default@FP_STEM_HP_WGS=100000 default@FP_STEM_HP_WGS=768 default@FP_STEM_HP_WGS=256 > $O/ab_step2.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps; FP_SERIAL=1 FP_PLAN=0 rocprofv3 --kernel-trace --stats -d /tmp/ps -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
python $R/scripts/rocprof_summary.py $(find /tmp/ps -name "*.db" | head -1) $O/kernel_stats_serial.txt "serial" > /dev/null 2>&1
cd $R
tail -2 $O/pytest_stem2.log; cat $O/ab_step2.txt; grep -i "stem\|pack_amax\|maxpool\|loss_kernel\|adam" $O/kernel_stats_serial.txt
