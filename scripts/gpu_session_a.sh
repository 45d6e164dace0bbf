#!/bin/bash
# round 2, GPU session A: hazard reproducer, the new parity / trainer / DP tests, a first bench line per workload
set +e
export TMPDIR=/tmp
O=gpurun_out/r2a
mkdir -p $O
( timeout 200 scripts/ubench/bin/pk_hazard 6 ) > $O/pk_hazard.txt 2>&1
echo "pk_hazard rc=$?"; tail -12 $O/pk_hazard.txt
( timeout 1500 python -m pytest tests/test_gpu_parity_fullsize.py tests/test_gpu_trainer.py tests/test_gpu_dp.py -q -m gpu -s --durations=8 ) > $O/pytest_new.log 2>&1
echo "pytest new rc=$?"; tail -40 $O/pytest_new.log
( timeout 400 python bench.py --dump-kernels $O/kernels_kitti.json ) > $O/bench_kitti.json 2> $O/bench_kitti.err
echo "bench kitti rc=$?"; tail -3 $O/bench_kitti.err; head -c 3000 $O/bench_kitti.json
( timeout 400 python bench.py --workload matterport --dump-kernels $O/kernels_mp.json ) > $O/bench_mp.json 2> $O/bench_mp.err
echo "bench mp rc=$?"; tail -3 $O/bench_mp.err; head -c 1500 $O/bench_mp.json
( timeout 300 python bench.py --force-dist --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events ) > $O/bench_forcedist.json 2> $O/bench_forcedist.err
echo "bench force-dist rc=$?"; tail -3 $O/bench_forcedist.err; head -c 600 $O/bench_forcedist.json
