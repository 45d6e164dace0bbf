#!/bin/bash
# round 2, GPU session B: hazard follow-ups, the 1x512x640 depth-decoder gradient anomaly, wgrad v2 A/B
set +e
export TMPDIR=/tmp
O=gpurun_out/r2b
mkdir -p $O
( timeout 300 scripts/ubench/bin/pk_hazard 6 ) > $O/pk_hazard.txt 2>&1
echo "pk_hazard rc=$?"; grep -E "slp|RESULT|alone" $O/pk_hazard.txt
( FP_LIB=$PWD/scripts/ubench/bin/lib_slp.so timeout 300 python scripts/debug_head_wgrad_det.py ) > $O/head_wgrad_det_slp.txt 2>&1
echo "head_wgrad det (SLP build) rc=$?"; grep -v "^ block" $O/head_wgrad_det_slp.txt | tail -14
( timeout 600 python scripts/debug_parity_stage.py 1 512 640 ) > $O/parity_1x512x640.txt 2>&1
echo "parity stage rc=$?"; grep -v "gpu .* cpu32 .*e-0[0-9]$" $O/parity_1x512x640.txt | head -60; grep -c "<<<" $O/parity_1x512x640.txt
( timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wgrad" ) > $O/pytest_wgrad.log 2>&1
echo "pytest wgrad rc=$?"; tail -5 $O/pytest_wgrad.log
( timeout 200 python scripts/conv_microbench.py wgrad 20 ) > $O/mb_wgrad_v2.txt 2>&1
( FP_WGRAD_BF3_V1=1 timeout 200 python scripts/conv_microbench.py wgrad 20 ) > $O/mb_wgrad_v1.txt 2>&1
echo "--- v2"; grep bf16x3 $O/mb_wgrad_v2.txt; echo "--- v1"; grep bf16x3 $O/mb_wgrad_v1.txt
for v in 0 1 0 1; do
  ( FP_WGRAD_BF3_V1=$v timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events ) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('V1=$v', d['value'], d['ms_per_step'], d['decoder_backward']['ms'])"
done
