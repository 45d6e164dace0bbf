#!/bin/bash
# round 4, session AE (branch next/ct-epilogue): transposed-accumulator epilogue of the tile kernel -- bit comparison + times + one step pair
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
O=gpurun_out/r4ae; mkdir -p $O
CT=$PWD/scripts/ubench/bin/lib_ct.so
TILE_HASH=1 TILE_SHAPES=0,1,4,7 timeout 40 python scripts/tile_bench.py default 20 2>/dev/null | grep -v amdgpu.ids > $O/tile.txt
FP_LIB=$CT FP_TILE_CT=1 TILE_HASH=1 TILE_SHAPES=0,1,4,7 timeout 40 python scripts/tile_bench.py ct 20 2>/dev/null | grep -v amdgpu.ids >> $O/tile.txt
cat $O/tile.txt
for spec in default ct; do
  lib=$PWD/footprints_amd/libfootprints_hip.so; e=0; [ $spec = ct ] && lib=$CT && e=1
  echo -n "$spec " >> $O/step.txt
  FP_LIB=$lib FP_TILE_CT=$e timeout 40 python bench.py --leg train-only --steps 30 --warmup 8 2>/dev/null | tail -1 >> $O/step.txt
done
cat $O/step.txt
