#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4u; mkdir -p $O
cd $R
: > $O/bisect2.txt
for combo in "tests/test_gpu_kernels.py tests/test_gpu_network.py" "tests/test_gpu_hp.py tests/test_gpu_network.py" "tests/test_gpu_data_path.py tests/test_gpu_dp.py tests/test_gpu_network.py"; do
  echo "== $combo" >> $O/bisect2.txt
  timeout 600 python -m pytest $combo -x -q 2>&1 | grep -E "passed|failed|AssertionError" | head -4 >> $O/bisect2.txt
done
cat $O/bisect2.txt
