#!/bin/bash
# GPU box, round 4 session D: planner thresholds re-measured on the faster kernels (environment knobs, no rebuild) + a timeline of the step
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r4d; mkdir -p $out; : > $out/summary.txt
t0=$(date +%s)
scripts/ab_lib_step.sh kitti rounds=2 default default@FP_TILE_SK1_FROM=256 default@FP_TILE_SK1_FROM=400 default@FP_TILE_WPF_MAX_WG=800 default@FP_WGRAD_TARGET_WGS=384 default@FP_BN_ROWS_PER_THREAD=8 default@FP_TILE_SK1_FROM=100 > $out/ab_step.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats -d /tmp/pc -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
DB=$(find /tmp/pc -name "*.db" | head -1)
python $R/scripts/rocprof_summary.py $DB $R/$out/kernel_stats_concurrent_kitti.txt "rocprofv3 --kernel-trace --stats -- python scripts/step_loop.py kitti 5 3" > /dev/null
python $R/scripts/timeline.py $DB > $R/$out/timeline_concurrent_step.txt 2>&1
rm -rf /tmp/ps; FP_SERIAL=1 FP_PLAN=0 rocprofv3 --kernel-trace --stats -d /tmp/ps -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
python $R/scripts/rocprof_summary.py $(find /tmp/ps -name "*.db" | head -1) $R/$out/kernel_stats_serial_kitti.txt "FP_SERIAL=1 FP_PLAN=0 rocprofv3 --kernel-trace --stats -- python scripts/step_loop.py kitti 5 3" > /dev/null
cd $R
echo "prof done t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
cat $out/summary.txt $out/ab_step.txt; head -50 $out/timeline_concurrent_step.txt
