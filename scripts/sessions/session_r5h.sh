#!/bin/bash
# GPU box: why did test_g2_decoder_golden_through_engine fail inside the whole suite (session G) and pass in sessions A, B?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5h; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 300 python -m pytest tests/test_gpu_network.py -q -p no:cacheprovider -k g2 > $O/alone.log 2>&1; echo "alone rc=$? t=$(( $(date +%s)-t0 ))" > $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py -q -p no:cacheprovider -k "not fullsize" > $O/after_kernels.log 2>&1; echo "after_kernels rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_hp.py tests/test_gpu_network.py -q -p no:cacheprovider > $O/after_hp.log 2>&1; echo "after_hp rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_golden_blocks.py tests/test_gpu_network.py -q -p no:cacheprovider > $O/after_blocks.log 2>&1; echo "after_blocks rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_data_path.py tests/test_gpu_dp.py tests/test_gpu_network.py -q -p no:cacheprovider > $O/after_dp.log 2>&1; echo "after_dp rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
FP_NEED32_SYNC=1 timeout 900 python -m pytest tests/test_gpu_data_path.py tests/test_gpu_dp.py tests/test_gpu_golden_blocks.py tests/test_gpu_hp.py tests/test_gpu_kernels.py tests/test_gpu_network.py -q -p no:cacheprovider > $O/all_sync.log 2>&1; echo "all_with_sync rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_data_path.py tests/test_gpu_dp.py tests/test_gpu_golden_blocks.py tests/test_gpu_hp.py tests/test_gpu_kernels.py tests/test_gpu_network.py -q -p no:cacheprovider > $O/all_nosync.log 2>&1; echo "all_no_sync rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
cat $O/summary.txt
for f in alone after_kernels after_hp after_blocks after_dp all_sync all_nosync; do echo "== $f"; grep -E "passed|failed" $O/$f.log | tail -1; grep "^FAILED\|AssertionError: dec" $O/$f.log | head -3; done
