#!/bin/bash
# GPU box, round 5 session B: the exact-format catch-up (weight gradient on the prefetch ring, BatchNorm backward sums out of the data gradient's
# epilogue, bf16x3 flattened GEMM) -- kernel tests of the new paths, alternating A/B of the training step, then the full-size parity cases in
# both formats under the decision-forced rule.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_hp.py tests/test_gpu_kernels.py tests/test_gpu_golden_blocks.py -q -p no:cacheprovider -x > $O/pytest_kernels.log 2>&1; echo "kernels rc=$? t=$(( $(date +%s)-t0 ))" > $O/summary.txt
tail -3 $O/pytest_kernels.log
timeout 600 python -m pytest tests/test_gpu_switches.py tests/test_gpu_network.py tests/test_gpu_trainer.py -q -p no:cacheprovider > $O/pytest_net.log 2>&1; echo "net rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
tail -3 $O/pytest_net.log
ab() { # tag, env...
  tag=$1; shift
  env "$@" timeout 120 python bench.py --leg train-only --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['img_per_s'], d['ms_per_step'], d['final_loss'])" >> $O/ab.txt 2>&1
}
for rep in 1 2; do
  ab r4_path FP_WGRAD_PF=0 FP_BN_BWD_EPI=0 FP_BF3_IGEMM=0
  ab ring FP_BN_BWD_EPI=0 FP_BF3_IGEMM=0
  ab ring_bnb FP_BF3_IGEMM=0
  ab all X=1
  ab all_wg384 FP_WGRAD_TARGET_WGS=384
  ab all_wg512 FP_WGRAD_TARGET_WGS=512
  ab all_wg192 FP_WGRAD_TARGET_WGS=192
done
ab all_sk1_96 FP_TILE_SK1_FROM=96
ab all_sk1_256 FP_TILE_SK1_FROM=256
ab all_layout0123 FP_STREAM_LAYOUT=0,1,2,0
ab pair FP_OPERANDS=fp16_pair
echo "ab done t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
cat $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py -q -p no:cacheprovider --durations=12 > $O/pytest_parity.log 2>&1; echo "parity rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
tail -25 $O/pytest_parity.log
cp -r gpurun_out/parity $O/ 2>/dev/null
cat $O/summary.txt
