#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out/r2e
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "wgrad" 2>&1 | tail -3
for sh in "64 64 96 320" "32 32 192 640" "64 64 48 160" "128 128 24 80" "256 256 12 40"; do
  python scripts/wgrad_one.py $sh 2>/dev/null | tail -1
  FP_WGRAD_BF3_V=2 python scripts/wgrad_one.py $sh 2>/dev/null | tail -1 | sed 's/^/   v2: /'
done
for v in 3 2 3 2; do
  ( FP_WGRAD_BF3_V=$v timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events ) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('V=$v', d['value'], d['ms_per_step'], d['decoder_backward']['ms'])"
done
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $P --output-format csv -d /root/repo/$O/pmc$i -o pmc -- python /root/repo/scripts/wgrad_one.py 64 64 96 320 12 5 ) > $O/pmc$i.log 2>&1
done
python scripts/pmc_sq.py wgrad3x3_bf3_v3 $O/pmc1 $O/pmc2
timeout 900 python -m pytest tests/test_gpu_network.py -q -m gpu -x -k "full_size or oracle" 2>&1 | tail -3
