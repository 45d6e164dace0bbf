#!/bin/bash
# GPU box: round 6's committed evidence from ONE build (the tree as it is), per OPERAND FORMAT (exact = default = `value`; fp16_pair = opt-in):
# HBM traffic per kernel family (two counters-only PMC passes, digest-stamped), step-level MFMA busy % and HBM bytes (scripts/pmc_step.py),
# rocprofv3 kernel stats on one stream, a timeline of the default schedule; then the bench lines (KITTI with both legs, Matterport, forced
# data-parallel branch) with the traffic of THIS build attached.  Outputs under gpurun_out/r6prof/ (copied to profiles/ by hand).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
t0=$(date +%s)
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"
for fmt in exact fp16_pair; do
  export FP_OPERANDS=$fmt
  wl=kitti
  python $R/bench.py --workload $wl --steps 20 --warmup 5 --sustain 0 --dump-kernels $O/kernels_${wl}_$fmt.json --no-cpu-baseline --no-other-format --no-loader > $O/bench_pre_${wl}_$fmt.json 2>/dev/null
  ms=$(python -c "import json,sys; print(json.loads(open('$O/bench_pre_${wl}_$fmt.json').read().strip().splitlines()[-1])['ms_per_step'])")
  rm -rf /tmp/pf /tmp/pw /tmp/pq /tmp/ps
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o pmc -- python $R/scripts/step_loop.py $wl 2 1 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o pmc -- python $R/scripts/step_loop.py $wl 2 1 > /dev/null 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d /tmp/pq -o pmc -- python $R/scripts/step_loop.py $wl 2 1 > /dev/null 2>&1
  python $R/scripts/pmc_hbm.py $wl /tmp/pf /tmp/pw $O/kernels_${wl}_$fmt.json $O/round6_pmc_hbm_${wl}_$fmt.json > /dev/null
  python $R/scripts/pmc_step.py $wl $fmt /tmp/pf /tmp/pw /tmp/pq 2 1 $O/round6_pmc_step_${wl}_$fmt.json $ms > /dev/null
  FP_SERIAL=1 FP_PLAN=0 rocprofv3 --kernel-trace --stats -d /tmp/ps -- python $R/scripts/step_loop.py $wl 5 3 > /dev/null 2>&1
  python $R/scripts/rocprof_summary.py $(find /tmp/ps -name "*.db" | head -1) $O/round6_kernel_stats_serial_${wl}_$fmt.txt "FP_OPERANDS=$fmt FP_SERIAL=1 FP_PLAN=0 rocprofv3 --kernel-trace --stats -- python scripts/step_loop.py $wl 5 3   (one stream: exclusive kernel durations; 8 train steps)"
  echo "$fmt passes done t=$(( $(date +%s)-t0 ))"
done
unset FP_OPERANDS
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats -d /tmp/pc -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
DB=$(find /tmp/pc -name "*.db" | head -1)
python $R/scripts/rocprof_summary.py $DB $O/round6_kernel_stats_concurrent_kitti_exact.txt "rocprofv3 --kernel-trace --stats -- python scripts/step_loop.py kitti 5 3   (default operand format and schedule: four hardware queues, recorded launch plan; 8 train steps)"
python $R/scripts/timeline.py $DB -2 trace > $O/round6_timeline_concurrent_step_exact.txt 2>&1
echo "timeline done t=$(( $(date +%s)-t0 ))"
cd $R
python scripts/hbm_microbench.py > $O/round6_hbm_kernels.txt 2>&1
cp $O/round6_pmc_hbm_*.json $O/round6_pmc_step_*.json $R/profiles/ 2>/dev/null       # bench.py attaches the counters of THIS build (digest checked)
# the stdout line is the compact record (round 6); the full record is bench_detail.json beside it -- both are kept
S=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round6_bench_line_kitti.json 2> $O/bench_kitti.err; cp bench_detail.json $O/round6_bench_detail_kitti.json
echo "driver-form bench wall: $(( $(date +%s)-S )) s" > $O/round6_bench_wall.txt
python bench.py --other-format --sustain 10 > $O/round6_bench_line_kitti_all_legs.json 2> $O/bench_kitti_all.err; cp bench_detail.json $O/round6_bench_detail_kitti_all_legs.json
python bench.py --workload matterport --no-cpu-baseline > $O/round6_bench_line_matterport.json 2> $O/bench_matterport.err; cp bench_detail.json $O/round6_bench_detail_matterport.json
python bench.py --force-dist --no-cpu-baseline --no-loader > $O/round6_bench_line_kitti_forced_dp.json 2>/dev/null; cp bench_detail.json $O/round6_bench_detail_kitti_forced_dp.json
echo "all done t=$(( $(date +%s)-t0 ))"
ls -la $O
