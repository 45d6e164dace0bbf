#!/bin/bash
# GPU box, end of a round: the statistics-sink kernel test (log kept), the profile session of the build, then the whole GPU suite (log kept)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 python -m pytest tests/test_gpu_hp.py -k "batchnorm_partials" -q > gpurun_out/final_bnpart.log 2>&1
tail -n 3 gpurun_out/final_bnpart.log
timeout 170 bash scripts/profile_session_r3.sh > gpurun_out/r3prof_session.log 2>&1
tail -n 2 gpurun_out/r3prof_session.log
timeout 330 python -m pytest tests/ -q -m gpu > gpurun_out/final_all.log 2>&1
tail -n 3 gpurun_out/final_all.log
