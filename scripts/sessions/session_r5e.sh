#!/bin/bash
# GPU box, round 5 session E: the ring weight gradients with their staging between the MFMAs as the default build (both operand formats), the
# pair-wise exact split (weight gradient, flattened GEMM): kernel tests, bit-identity switches, step A/B against -DFP_W3_INTERLEAVE=0 and the
# early-issue variant (=4).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5e; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_hp.py tests/test_gpu_kernels.py -q -p no:cacheprovider -k "wgrad or igemm" > $O/pytest_kernels.log 2>&1; echo "kernels rc=$? t=$(( $(date +%s)-t0 ))" > $O/summary.txt
tail -3 $O/pytest_kernels.log
timeout 600 python -m pytest tests/test_gpu_switches.py tests/test_gpu_trainer.py -q -p no:cacheprovider > $O/pytest_sw.log 2>&1; echo "switches rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
tail -3 $O/pytest_sw.log
bash scripts/ab_lib_step.sh kitti rounds=2 il0 default il4 il0@FP_OPERANDS=fp16_pair default@FP_OPERANDS=fp16_pair > /dev/null 2>&1
cp gpurun_out/ab/step_kitti.txt $O/step_ab.txt; cat $O/step_ab.txt
echo "ab done t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
cat $O/summary.txt
