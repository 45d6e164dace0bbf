#!/bin/bash
# GPU box, end of a round: the whole GPU suite (log kept), then the profile session of the same build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_hp.py -k "batchnorm_partials" -q > gpurun_out/final_bnpart.log 2>&1
tail -n 3 gpurun_out/final_bnpart.log
python -m pytest tests/ -q -m gpu > gpurun_out/final_all.log 2>&1
tail -n 3 gpurun_out/final_all.log
bash scripts/profile_session_r3.sh > gpurun_out/r3prof_session.log 2>&1
tail -n 3 gpurun_out/r3prof_session.log
