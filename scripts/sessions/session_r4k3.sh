#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; mkdir -p $O
cd $R
: > $O/hazard3.txt
for v in hwfma4 hwfma5; do
  echo "== $v" >> $O/hazard3.txt
  FP_LIB=$R/scripts/ubench/bin/lib_$v.so timeout 200 python -m pytest tests/test_gpu_kernels.py -x -q -k "head_wgrad or head_backward or head" 2>&1 | tail -2 >> $O/hazard3.txt
  FP_LIB=$R/scripts/ubench/bin/lib_$v.so timeout 300 python scripts/debug_head_wgrad_det.py 2>&1 | grep -v amdgpu.ids >> $O/hazard3.txt
done
cat $O/hazard3.txt
