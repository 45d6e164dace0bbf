#!/bin/bash
# GPU box: how many __amd_rocclr_copyBuffer dispatches belong to a training step?  Same process with 5 and with 15 recorded steps:
# the difference / 10 is the per-step count, the rest is start-up (parameter flattening, state uploads).  -> gpurun_out/copybuffer.txt
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/copybuffer.txt
for n in 5 15; do
  rm -rf /tmp/cb$n
  rocprofv3 --kernel-trace --stats -d /tmp/cb$n -- python $R/scripts/step_loop.py kitti $n 3 > /dev/null 2>&1
  python $R/scripts/rocprof_summary.py $(find /tmp/cb$n -name "*.db" | head -1) /tmp/cb$n.txt "steps=$n" > /dev/null
  echo "train steps: $((n + 3))  $(grep copyBuffer /tmp/cb$n.txt)" >> $R/gpurun_out/copybuffer.txt
done
cat $R/gpurun_out/copybuffer.txt
