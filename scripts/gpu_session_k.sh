#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out/r2k
mkdir -p $O
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P3="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES"
for sh in "64 64 96 320" "64 64 48 160" "256 256 12 40"; do
  python scripts/tile_one.py $sh 12 20 fwd 2>/dev/null | tail -1
  tag=$(echo $sh | tr ' ' '_')
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    ( cd /tmp && timeout 300 rocprofv3 --pmc $P --output-format csv -d /root/repo/$O/pmc_${tag}_$i -o pmc -- python /root/repo/scripts/tile_one.py $sh 12 5 fwd ) > $O/pmc_${tag}_$i.log 2>&1
  done
  python scripts/pmc_sq.py conv3x3_tile_bf3 $O/pmc_${tag}_1 $O/pmc_${tag}_2 $O/pmc_${tag}_3
done
