#!/bin/bash
# GPU box: SQ counters (three counters-only rocprofv3 passes each) of the exact-format tile kernel on the full-resolution 32-channel layers
# (the least efficient big launches of the step: 35 % of the MFMA peak) next to the 64-channel reference point -> gpurun_out/r6pmc/round6_pmc_sq_exact32.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P3="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
out=$O/round6_pmc_sq_exact32.txt
echo "# rocprofv3 --pmc <SQ counters, three counters-only passes> --output-format csv -- <microbench>; averages per launch (scripts/pmc_sq.py)" > $out
echo "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md)" >> $out
run() { # tag, kernel-name fragment, command...
  tag=$1; frag=$2; shift 2
  rm -rf /tmp/q1 /tmp/q2 /tmp/q3
  rocprofv3 --pmc $P1 --output-format csv -d /tmp/q1 -o pmc -- "$@" > /dev/null 2>&1
  rocprofv3 --pmc $P2 --output-format csv -d /tmp/q2 -o pmc -- "$@" > /dev/null 2>&1
  rocprofv3 --pmc $P3 --output-format csv -d /tmp/q3 -o pmc -- "$@" > /dev/null 2>&1
  echo "## $tag   ($*)" >> $out
  python $R/scripts/pmc_sq.py "$frag" /tmp/q1 /tmp/q2 /tmp/q3 >> $out
}
python $R/scripts/tile_one.py 32 32 192 640 12 10 fwd >> $out
python $R/scripts/tile_one.py 32 32 192 640 12 10 dgrad >> $out
python $R/scripts/tile_one.py 64 64 96 320 12 10 fwd >> $out
run "exact tile kernel, forward 32 -> 32 @ 192 x 640 x 12" "conv3x3_tile_bf3_kernel<8, 16, 32, 4, 1, false, false, 3, false, false>" python $R/scripts/tile_one.py 32 32 192 640 12 5 fwd
run "exact tile kernel, data gradient (fold) 32 -> 32 @ 192 x 640 x 12" "conv3x3_tile_bf3_kernel<8, 16, 32, 4, 1, true, true, 3, false, false>" python $R/scripts/tile_one.py 32 32 192 640 12 5 dgrad
run "exact tile kernel, forward 64 -> 64 @ 96 x 320 x 12" "conv3x3_tile_bf3_kernel<8, 16, 64, 2, 2, false, false, 3, false, false>" python $R/scripts/tile_one.py 64 64 96 320 12 5 fwd
cat $out
