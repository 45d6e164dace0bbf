#!/bin/bash
# GPU box: SQ counters (three counters-only rocprofv3 passes each) of the fp16-pair tile kernel (forward and data gradient), the fp16-pair
# weight-gradient kernel and the flattened fp16-pair kernel on their microbenchmarks -> gpurun_out/r3pmc/round3_pmc_sq_hp.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P3="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
out=$O/round3_pmc_sq_hp.txt
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
run "fp16-pair tile kernel, forward 64 -> 64 @ 96 x 320 x 12" "conv3x3_tile_bf3_kernel<8, 16, 64, 2, 2, false, false, 2, true>" python $R/scripts/hp_one.py 64 64 96 320 12 5 fwd
run "fp16-pair tile kernel, data gradient (reflection fold) 64 -> 64 @ 96 x 320 x 12" "conv3x3_tile_bf3_kernel<8, 16, 64, 2, 2, true, true, 2, true>" python $R/scripts/hp_one.py 64 64 96 320 12 5 dgrad
run "fp16-pair tile kernel, forward 256 -> 256 @ 12 x 40 x 12 (6 x 20 tiles)" "conv3x3_tile_bf3_kernel<6, 20, 64, 2, 2, false, false, 2, true>" python $R/scripts/hp_one.py 256 256 12 40 12 5 fwd
run "fp16-pair weight gradient (third generation) 64 -> 64 @ 96 x 320 x 12" "wgrad3x3_bf3_v3_kernel<1, 2>" python $R/scripts/wgrad_one.py 64 64 96 320 12 5
cat $out | head -60
