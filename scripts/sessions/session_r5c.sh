#!/bin/bash
# GPU box, round 5 session C: kernel tests of the exact-format paths again (tolerance of the BatchNorm-sum test), the data-parallel tests with the
# new shared-GPU opt-in and the forced bench line, and planner knobs re-measured for the EXACT operand format (rounds 2-4 tuned them on fp16 pairs).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_hp.py tests/test_gpu_kernels.py -q -p no:cacheprovider > $O/pytest_kernels.log 2>&1; echo "kernels rc=$? t=$(( $(date +%s)-t0 ))" > $O/summary.txt
tail -3 $O/pytest_kernels.log
timeout 900 python -m pytest tests/test_gpu_dp.py -q -p no:cacheprovider > $O/pytest_dp.log 2>&1; echo "dp rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
tail -5 $O/pytest_dp.log
ab() { # tag, env...
  tag=$1; shift
  env "$@" timeout 120 python bench.py --leg train-only --steps 30 --warmup 8 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['img_per_s'], d['ms_per_step'], d['final_loss'])" >> $O/ab.txt 2>&1
}
for rep in 1 2; do
  ab base X=1
  ab bn32_400 FP_TILE_BN32_BELOW=400
  ab bn32_800 FP_TILE_BN32_BELOW=800
  ab sktarget512 FP_TILE_SK_TARGET=512
  ab pwgrad512 FP_PWGRAD_TARGET_WGS=512
  ab pwgrad1024 FP_PWGRAD_TARGET_WGS=1024
  ab igemm_sk1_96 FP_IGEMM_SK1_FROM=96
  ab igemm_sk1_320 FP_IGEMM_SK1_FROM=320
done
ab maxsk8 FP_TILE_MAX_SK=8
ab bnrows8 FP_BN_ROWS_PER_THREAD=8
ab bnrows2 FP_BN_ROWS_PER_THREAD=2
ab layout0122_dsaux0 FP_DS_AUX=0
EXTRA='--workload matterport' ab mp_base X=1
echo "ab done t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
cat $O/ab.txt
cat $O/summary.txt
