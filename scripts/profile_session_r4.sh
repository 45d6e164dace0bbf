#!/bin/bash
# GPU box: the round's committed evidence from ONE build (the tree as it is) -- bench lines (both workloads + the forced data-parallel branch),
# rocprofv3 kernel stats (one stream and the default schedule), HBM traffic per kernel family (two counters-only PMC passes, stamped with the
# digest of the kernel source they were measured on), SQ counters of the tile kernel (large grid, small-grid WPF) and the weight gradient, a
# steady-state timeline, the HBM-bound kernels' table.  Outputs under gpurun_out/r4prof/ (copied to profiles/ by hand).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
t0=$(date +%s)
for wl in kitti matterport; do
  python $R/bench.py --workload $wl --steps 20 --warmup 5 --dump-kernels $O/kernels_$wl.json --no-cpu-baseline --no-exact-split --no-loader > $O/bench_pre_$wl.json 2>/dev/null
  rm -rf /tmp/pf /tmp/pw /tmp/ps
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o pmc -- python $R/scripts/step_loop.py $wl 2 1 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o pmc -- python $R/scripts/step_loop.py $wl 2 1 > /dev/null 2>&1
  python $R/scripts/pmc_hbm.py $wl /tmp/pf /tmp/pw $O/kernels_$wl.json $O/round4_pmc_hbm_$wl.json > /dev/null
  FP_SERIAL=1 FP_PLAN=0 rocprofv3 --kernel-trace --stats -d /tmp/ps -- python $R/scripts/step_loop.py $wl 5 3 > /dev/null 2>&1
  python $R/scripts/rocprof_summary.py $(find /tmp/ps -name "*.db" | head -1) $O/round4_kernel_stats_serial_$wl.txt "FP_SERIAL=1 FP_PLAN=0 rocprofv3 --kernel-trace --stats -- python scripts/step_loop.py $wl 5 3   (one stream: exclusive kernel durations; 8 train steps)"
  echo "$wl passes done t=$(( $(date +%s)-t0 ))"
done
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats -d /tmp/pc -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
DB=$(find /tmp/pc -name "*.db" | head -1)
python $R/scripts/rocprof_summary.py $DB $O/round4_kernel_stats_concurrent_kitti.txt "rocprofv3 --kernel-trace --stats -- python scripts/step_loop.py kitti 5 3   (default schedule: four hardware queues, recorded launch plan; 8 train steps)"
python $R/scripts/timeline.py $DB -2 trace > $O/round4_timeline_concurrent_step.txt 2>&1
echo "timeline done t=$(( $(date +%s)-t0 ))"
# ---- SQ counters: three counters-only passes per kernel on its microbenchmark
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P3="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
out=$O/round4_pmc_sq_hp.txt
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
run "fp16-pair tile kernel, forward 64 -> 64 @ 96 x 320 x 12 (large grid)" "conv3x3_tile_bf3_kernel<8, 16, 64, 2, 2, false, false, 2, true, false>" python $R/scripts/hp_one.py 64 64 96 320 12 5 fwd
run "fp16-pair tile kernel, data gradient (reflection fold) 64 -> 64 @ 96 x 320 x 12" "conv3x3_tile_bf3_kernel<8, 16, 64, 2, 2, true, true, 2, true, false>" python $R/scripts/hp_one.py 64 64 96 320 12 5 dgrad
run "fp16-pair tile kernel, forward 256 -> 256 @ 12 x 40 x 12 (6 x 20 tiles, small-grid WPF variant)" "conv3x3_tile_bf3_kernel<6, 20, 64, 2, 2, false, false, 2, true, true>" python $R/scripts/hp_one.py 256 256 12 40 12 5 fwd
run "fp16-pair weight gradient (prefetch ring) 64 -> 64 @ 96 x 320 x 12" "wgrad3x3_hp_pf_kernel<1, 2>" python $R/scripts/wgrad_one.py 64 64 96 320 12 5
echo "sq done t=$(( $(date +%s)-t0 ))"
cd $R
python scripts/hbm_microbench.py > $O/round4_hbm_kernels.txt 2>&1
python scripts/tile_bench.py default 30 > $O/round4_tile_bench.txt 2>&1
cp $O/round4_pmc_hbm_*.json $R/profiles/ 2>/dev/null       # bench.py attaches the traffic of THIS build (digest checked)
for wl in kitti matterport; do python bench.py --workload $wl > $O/round4_bench_line_$wl.json 2> $O/bench_$wl.err; done
python bench.py --force-dist --no-cpu-baseline --no-loader > $O/round4_bench_line_kitti_forced_dp.json 2>/dev/null
echo "all done t=$(( $(date +%s)-t0 ))"
ls -la $O
