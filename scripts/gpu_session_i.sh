#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out/r2i
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_dp.py tests/test_gpu_trainer.py -q -m gpu -x ) 2>&1 | tail -12
for v in 1 0 1 0; do
  ( FP_PLAN=$v timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-events ) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PLAN=$v', d['value'], d['ms_per_step'], d['step_ms']['median'], d['device_data_path']['ms_per_step'])"
done
python scripts/host_overhead.py 2>&1 | tail -4
