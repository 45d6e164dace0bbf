#!/bin/bash
# GPU box, final build of round 4: the whole GPU suite + smoke, then the round's profile set from the same build
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r4v; mkdir -p $out; : > $out/summary.txt
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_all.log 2>&1; echo "pytest all rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$? t=$(( $(date +%s)-t0 ))" >> $out/summary.txt
cat $out/summary.txt; tail -5 $out/pytest_all.log; tail -2 $out/smoke.log
bash scripts/profile_session_r4.sh > $out/profile_session.log 2>&1; echo "profile rc=$? t=$(( $(date +%s)-t0 ))"
tail -5 $out/profile_session.log
