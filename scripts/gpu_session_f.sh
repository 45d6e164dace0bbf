#!/bin/bash
set +e
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wgrad" 2>&1 | tail -3
for sh in "64 64 96 320" "32 32 192 640" "64 64 48 160" "128 128 24 80" "256 256 12 40"; do
  python scripts/wgrad_one.py $sh 2>/dev/null | tail -1
  FP_WGRAD_SPLIT_REDUCE=1 python scripts/wgrad_one.py $sh 2>/dev/null | tail -1 | sed 's/^/   split-reduce: /'
  FP_WGRAD_BF3_V=2 python scripts/wgrad_one.py $sh 2>/dev/null | tail -1 | sed 's/^/   v2: /'
done
FP_W3_STAMPS=/tmp/st.txt python scripts/wgrad_one.py 64 64 96 320 12 3 2>/dev/null | tail -1; python scripts/stamps_summary.py /tmp/st.txt
for v in 3 2 3 2; do
  ( FP_WGRAD_BF3_V=$v timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events ) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('V=$v', d['value'], d['ms_per_step'], d['decoder_backward']['ms'])"
done
timeout 900 python -m pytest tests/test_gpu_network.py -q -m gpu -x -k "full_size or oracle" 2>&1 | tail -3
