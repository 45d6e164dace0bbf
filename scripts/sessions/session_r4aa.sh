#!/bin/bash
# round 4, session AA: Adam in pieces under the backward pass (FP_ADAM_STAGED, default on): exact switch test, step A/B, network tests
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
O=gpurun_out/r4aa; mkdir -p $O
t0=$(date +%s)
timeout 200 python -m pytest tests/test_gpu_switches.py -q -k "env11" > $O/pytest_switch.log 2>&1; echo "switch rc=$? t=$(( $(date +%s)-t0 ))"; tail -3 $O/pytest_switch.log
bash scripts/ab_lib_step.sh kitti rounds=3 default default@FP_ADAM_STAGED=0 > $O/step.txt 2>&1; tail -6 $O/step.txt
echo "t=$(( $(date +%s)-t0 ))"
timeout 300 python -m pytest tests/test_gpu_network.py tests/test_gpu_trainer.py -q -x > $O/pytest_net.log 2>&1; echo "net rc=$? t=$(( $(date +%s)-t0 ))"; tail -3 $O/pytest_net.log
