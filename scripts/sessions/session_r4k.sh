#!/bin/bash
# GPU box: round 1's run-to-run differences of head_wgrad next to the split-operand tile convolution, on the scatter-form kernel in three builds
# (scalar v_fma_f32 = the library; v_pk_fma_f32 with op_sel broadcast; v_pk_fma_f32 on materialised pairs)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "head" > $O/pytest_head.log 2>&1
echo "pytest head rc=$? t=$(( $(date +%s)-t0 ))"
timeout 300 python scripts/hbm_microbench.py > $O/hbm_kernels.txt 2>&1
for v in default hwfma1 hwfma2; do
  lib=$R/footprints_amd/libfootprints_hip.so
  [ $v != default ] && lib=$R/scripts/ubench/bin/lib_$v.so
  echo "== $v" >> $O/hazard.txt
  FP_LIB=$lib timeout 300 python scripts/debug_head_wgrad_det.py >> $O/hazard.txt 2>&1
  echo "hazard $v done t=$(( $(date +%s)-t0 ))"
done
tail -3 $O/pytest_head.log; grep -i "head_wgrad" $O/hbm_kernels.txt; cat $O/hazard.txt
