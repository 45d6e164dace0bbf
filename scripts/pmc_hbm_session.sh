#!/bin/bash
# GPU box: HBM traffic per kernel family (two counters-only PMC passes per workload) + the bench lines that carry it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in kitti matterport; do
  python $R/bench.py --workload $wl --dump-kernels $O/kernels_$wl.json --no-cpu-baseline --no-exact-split --no-loader > /dev/null 2>&1
  rm -rf /tmp/pf /tmp/pw
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o pmc -- python $R/scripts/step_loop.py $wl 2 1 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o pmc -- python $R/scripts/step_loop.py $wl 2 1 > /dev/null 2>&1
  python $R/scripts/pmc_hbm.py $wl /tmp/pf /tmp/pw $O/kernels_$wl.json $O/round3_pmc_hbm_$wl.json > /dev/null
  cp $O/round3_pmc_hbm_$wl.json $R/profiles/
done
cd $R
for wl in kitti matterport; do python bench.py --workload $wl > $O/round3_bench_line_$wl.json 2> $O/bench_$wl.err; done
python bench.py --force-dist --no-cpu-baseline --no-loader --no-kernel-events > $O/round3_bench_line_kitti_forced_dp.json 2>/dev/null
