#!/bin/bash
# GPU box, round 4 session A: the rewritten tile kernel (buffer loads; persistent / 16x16-tile options) against round 3's
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r4a; mkdir -p $out; : > $out/summary.txt
B=$PWD/scripts/ubench/bin
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_hp.py tests/test_gpu_kernels.py tests/test_gpu_switches.py -x -q > $out/pytest_default.log 2>&1; echo "pytest default rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
for v in persist3 persist4; do
  FP_LIB=$B/lib_$v.so timeout 400 python -m pytest tests/test_gpu_hp.py -x -q -k "conv3x3_hp" > $out/pytest_$v.log 2>&1; echo "pytest $v rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
done
FP_TILE_BIG_MIN_WG=1 timeout 400 python -m pytest tests/test_gpu_hp.py -x -q -k "conv3x3_hp" > $out/pytest_big.log 2>&1; echo "pytest big rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
: > $out/tile_bench.txt
for v in r3 default persist3 persist4; do
  lib=$B/lib_$v.so; [ $v = default ] && lib=$PWD/footprints_amd/libfootprints_hip.so
  FP_LIB=$lib timeout 300 python scripts/tile_bench.py $v >> $out/tile_bench.txt 2>&1
done
FP_TILE_BIG_MIN_WG=512 timeout 300 python scripts/tile_bench.py big512 >> $out/tile_bench.txt 2>&1
FP_TILE_PERSIST=0 FP_LIB=$B/lib_persist3.so TILE_SHAPES=0,1,4,5 timeout 300 python scripts/tile_bench.py persist3_off >> $out/tile_bench.txt 2>&1
echo "bench done t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
scripts/ab_lib_step.sh kitti rounds=2 r3 default persist3 default@FP_TILE_BIG_MIN_WG=512 persist3@FP_TILE_BIG_MIN_WG=512 > $out/ab_step.txt 2>&1
echo "ab done t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
cat $out/summary.txt; cat $out/tile_bench.txt; cat $out/ab_step.txt; tail -3 $out/pytest_*.log
