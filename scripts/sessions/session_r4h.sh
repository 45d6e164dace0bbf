#!/bin/bash
# GPU box: (1) the last-BatchNorm error decomposition inside the 12x192x640 parity case; (2) an ordered kernel trace of one steady-state step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
t0=$(date +%s)
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats -d /tmp/pc -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
DB=$(find /tmp/pc -name "*.db" | head -1)
python $R/scripts/timeline.py $DB -2 trace > $O/trace_step.txt 2>&1
echo "trace done t=$(( $(date +%s)-t0 ))"
cd $R
timeout 900 python -m pytest "tests/test_gpu_parity_fullsize.py::test_train_step_fp64_anchored[12-192-640]" -x -q -s > $O/pytest_decomp.log 2>&1
echo "pytest rc=$? t=$(( $(date +%s)-t0 ))"
cat gpurun_out/parity/last_bn_decomposition.json
tail -5 $O/pytest_decomp.log
