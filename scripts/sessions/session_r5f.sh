#!/bin/bash
# GPU box, round 5 session F: persistent workgroups for the EXACT tile kernel's large grids (-DFP_TILE_PERSIST_BUILD=2: they run three workgroups
# per CU anyway, so the resident loop's registers cost no occupancy) -- kernel tests on the variant, microbenchmarks, training step A/B.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O
cd $R
t0=$(date +%s)
P2=$R/scripts/ubench/bin/lib_p2.so
FP_LIB=$P2 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_hp.py -q -p no:cacheprovider -k "bf3 or tile or hp_kernel or emits" > $O/pytest_kernels.log 2>&1; echo "kernels(p2) rc=$? t=$(( $(date +%s)-t0 ))" > $O/summary.txt
tail -3 $O/pytest_kernels.log
FP_LIB=$P2 timeout 600 python -m pytest tests/test_gpu_network.py -q -p no:cacheprovider > $O/pytest_net.log 2>&1; echo "net(p2) rc=$? t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
tail -3 $O/pytest_net.log
for lib in default p2; do
  L=$R/footprints_amd/libfootprints_hip.so; [ $lib != default ] && L=$P2
  for shape in "64 64 96 320" "32 32 192 640" "64 64 48 160"; do
    for mode in fwd dgrad; do
      echo -n "$lib " >> $O/tile_ubench.txt
      FP_LIB=$L python scripts/hp_one.py $shape 12 20 $mode 2>&1 | tail -1 >> $O/tile_ubench.txt
    done
  done
done
cat $O/tile_ubench.txt
bash scripts/ab_lib_step.sh kitti rounds=2 default p2 p2@FP_TILE_PERSIST=0 > /dev/null 2>&1
cp gpurun_out/ab/step_kitti.txt $O/step_ab.txt; cat $O/step_ab.txt
bash scripts/ab_lib_step.sh matterport rounds=1 default p2 > /dev/null 2>&1
cp gpurun_out/ab/step_matterport.txt $O/step_ab_mp.txt; cat $O/step_ab_mp.txt
echo "ab done t=$(( $(date +%s)-t0 ))" >> $O/summary.txt
cat $O/summary.txt
