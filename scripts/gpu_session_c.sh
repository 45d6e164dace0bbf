#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out/r2c
mkdir -p $O
for sh in "64 64 96 320" "32 32 192 640" "64 64 48 160" "128 128 24 80" "256 256 12 40"; do
  python scripts/wgrad_one.py $sh 2>/dev/null | tail -1
  FP_WGRAD_NO_XCD=1 python scripts/wgrad_one.py $sh 2>/dev/null | tail -1 | sed 's/^/   no-xcd: /'
done
rocprofv3 -L > $O/counters.txt 2>&1
grep -c "SQ_" $O/counters.txt
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P3="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $P --output-format csv -d /root/repo/$O/pmc$i -o pmc -- python /root/repo/scripts/wgrad_one.py 64 64 96 320 12 5 ) > $O/pmc$i.log 2>&1
  echo "pass $i rc=$?"; tail -2 $O/pmc$i.log
done
python scripts/pmc_sq.py wgrad3x3_bf3_v2 $O/pmc1 $O/pmc2 $O/pmc3
timeout 1500 python -m pytest tests/test_gpu_parity_fullsize.py tests/test_gpu_trainer.py -q -m gpu -s > $O/pytest_parity.log 2>&1
echo "pytest rc=$?"; grep -E "worst|median|removed|passed|failed|Error" $O/pytest_parity.log | head -30
