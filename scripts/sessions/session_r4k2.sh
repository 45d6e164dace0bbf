#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; mkdir -p $O
cd $R
: > $O/hazard2.txt
for v in hwold_slp hwold_noslp hwfma3 hwred hwfma3red; do
  echo "== $v" >> $O/hazard2.txt
  FP_LIB=$R/scripts/ubench/bin/lib_$v.so timeout 300 python scripts/debug_head_wgrad_det.py 2>&1 | grep -v amdgpu.ids >> $O/hazard2.txt
done
cat $O/hazard2.txt
