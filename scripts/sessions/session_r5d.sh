#!/bin/bash
# GPU box, round 5 session D: where do the exact-format kernels wait?  SQ counters of the ring weight gradient and the tile kernel (exact
# operands), and the weight gradient with its staging between the MFMAs (library variants il1 / il2: -DFP_W3_INTERLEAVE=1 / 2) -- microbenchmarks,
# counters, training step.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
t0=$(date +%s)
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P3="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
out=$O/round5_pmc_sq_exact.txt
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
export WG_HP=0
run "exact ring weight gradient 64 -> 64 @ 96 x 320 x 12 (default build)" "wgrad3x3_hp_pf_kernel<1, 2, 3>" python $R/scripts/wgrad_one.py 64 64 96 320 12 5
FP_LIB=$R/scripts/ubench/bin/lib_il1.so run "exact ring weight gradient 64 -> 64 @ 96 x 320 x 12, staging between the MFMAs (il1)" "wgrad3x3_hp_pf_kernel<1, 2, 3>" env FP_LIB=$R/scripts/ubench/bin/lib_il1.so python $R/scripts/wgrad_one.py 64 64 96 320 12 5
run "exact ring weight gradient 256 -> 256 @ 12 x 40 x 12 (default build)" "wgrad3x3_hp_pf_kernel<1, 2, 3>" python $R/scripts/wgrad_one.py 256 256 12 40 12 5
run "exact tile kernel, forward 64 -> 64 @ 96 x 320 x 12 (large grid)" "conv3x3_tile_bf3_kernel<8, 16, 64, 2, 2, false, false, 3, false, false>" python $R/scripts/hp_one.py 64 64 96 320 12 5 fwd
run "exact tile kernel, forward 256 -> 256 @ 12 x 40 x 12 (6 x 20 tiles, 192 workgroups)" "conv3x3_tile_bf3_kernel<6, 20, 64, 2, 2, false, false, 3, false, false>" python $R/scripts/hp_one.py 256 256 12 40 12 5 fwd
echo "sq done t=$(( $(date +%s)-t0 ))" > $O/summary.txt
cd $R
for lib in default il1 il2; do
  L=$R/footprints_amd/libfootprints_hip.so; [ $lib != default ] && L=$R/scripts/ubench/bin/lib_$lib.so
  for shape in "64 64 96 320" "128 128 24 80" "256 256 12 40" "32 32 192 640" "512 512 6 20"; do
    echo -n "$lib " >> $O/wgrad_ubench.txt
    FP_LIB=$L WG_HP=0 python scripts/wgrad_one.py $shape 12 20 2>&1 | tail -1 >> $O/wgrad_ubench.txt
  done
done
echo "ubench done t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
cat $O/wgrad_ubench.txt
unset WG_HP
bash scripts/ab_lib_step.sh kitti rounds=2 default il1 il2 > /dev/null 2>&1
cp gpurun_out/ab/step_kitti.txt $O/step_ab.txt; cat $O/step_ab.txt
echo "ab done t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
cat $O/summary.txt; cat $out
