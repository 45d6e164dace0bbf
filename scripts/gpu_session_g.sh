#!/bin/bash
set +e
export TMPDIR=/tmp
for v in 3 2 3 2; do
  ( FP_SERIAL=1 FP_WGRAD_BF3_V=$v timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events ) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial V=$v', d['value'], d['ms_per_step'], d['decoder_backward']['ms'])"
done
