#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4w; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_hp.py tests/test_gpu_kernels.py tests/test_gpu_switches.py -q -k "wgrad or switch or Switch or env" > $O/pytest_wgrad.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_wgrad.log
