#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out/r2d
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wgrad" 2>&1 | tail -3
for sh in "64 64 96 320" "32 32 192 640" "64 64 48 160" "128 128 24 80" "256 256 12 40"; do
  python scripts/wgrad_one.py $sh 2>/dev/null | tail -1
  FP_WGRAD_NO_FAST=1 python scripts/wgrad_one.py $sh 2>/dev/null | tail -1 | sed 's/^/   no-fast: /'
done
for v in 0 1 0 1; do
  ( FP_WGRAD_NO_FAST=$v timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events ) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NO_FAST=$v', d['value'], d['ms_per_step'], d['decoder_backward']['ms'])"
done
timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_network.py -q -m gpu -x 2>&1 | tail -5
