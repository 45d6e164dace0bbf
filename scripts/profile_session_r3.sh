#!/bin/bash
# GPU box: the round's committed evidence from ONE build -- bench lines (both workloads), rocprofv3 kernel stats (one stream and the default
# schedule), HBM traffic per kernel family (two PMC passes), a steady-state timeline.  Outputs under gpurun_out/r3prof/ (copy to profiles/).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in kitti matterport; do
  python $R/bench.py --workload $wl --dump-kernels $O/kernels_$wl.json --no-cpu-baseline --no-exact-split --no-loader > $O/bench_pre_$wl.json 2>/dev/null
  rm -rf /tmp/pf /tmp/pw /tmp/ps
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o pmc -- python $R/scripts/step_loop.py $wl 2 1 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o pmc -- python $R/scripts/step_loop.py $wl 2 1 > /dev/null 2>&1
  python $R/scripts/pmc_hbm.py $wl /tmp/pf /tmp/pw $O/kernels_$wl.json $O/round3_pmc_hbm_$wl.json > /dev/null
  FP_SERIAL=1 FP_PLAN=0 rocprofv3 --kernel-trace --stats -d /tmp/ps -- python $R/scripts/step_loop.py $wl 5 3 > /dev/null 2>&1
  python $R/scripts/rocprof_summary.py $(find /tmp/ps -name "*.db" | head -1) $O/round3_kernel_stats_serial_$wl.txt "FP_SERIAL=1 FP_PLAN=0 rocprofv3 --kernel-trace --stats -- python scripts/step_loop.py $wl 5 3   (one stream: exclusive kernel durations; 8 train steps)"
done
rm -rf /tmp/pc; rocprofv3 --kernel-trace --stats -d /tmp/pc -- python $R/scripts/step_loop.py kitti 5 3 > /dev/null 2>&1
DB=$(find /tmp/pc -name "*.db" | head -1)
python $R/scripts/rocprof_summary.py $DB $O/round3_kernel_stats_concurrent_kitti.txt "rocprofv3 --kernel-trace --stats -- python scripts/step_loop.py kitti 5 3   (default schedule: four hardware queues, recorded launch plan; 8 train steps)"
python $R/scripts/timeline.py $DB > $O/round3_timeline_concurrent_step.txt 2>&1
cp $O/round3_pmc_hbm_*.json $R/profiles/ 2>/dev/null
cd $R
for wl in kitti matterport; do python bench.py --workload $wl > $O/round3_bench_line_$wl.json 2> $O/bench_$wl.err; done
python bench.py --force-dist --no-cpu-baseline --no-loader --no-kernel-events > $O/round3_bench_line_kitti_forced_dp.json 2>/dev/null
ls -la $O
