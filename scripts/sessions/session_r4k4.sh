#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; mkdir -p $O
cd $R
: > $O/opsel_probe.txt
for v in hwfma4 hwfma6 hwfma3; do
  echo "== $v" >> $O/opsel_probe.txt
  FP_LIB=$R/scripts/ubench/bin/lib_$v.so timeout 200 python scripts/pk_opsel_probe.py 2>&1 | grep -v amdgpu.ids >> $O/opsel_probe.txt
done
cat $O/opsel_probe.txt
