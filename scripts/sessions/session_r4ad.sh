#!/bin/bash
# round 4, session AD: staged Adam with fewer workgroups per piece (FP_ADAM_MAX_WGS): step A/B on one box
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
O=gpurun_out/r4ad; mkdir -p $O
bash scripts/ab_lib_step.sh kitti rounds=2 default default@FP_ADAM_STAGED=1 default@FP_ADAM_STAGED=1,FP_ADAM_MAX_WGS=512 default@FP_ADAM_STAGED=1,FP_ADAM_MAX_WGS=256 > $O/step.txt 2>&1; tail -8 $O/step.txt
