#!/bin/bash
# GPU box: training-step A/B of library builds (scripts/build_variant.sh), alternating runs.
#   scripts/ab_lib_step.sh [workload] [rounds=N] <name|default>[@VAR=val[,VAR=val]] ...     results: gpurun_out/ab/step_<workload>.txt
set -u
cd "$(dirname "$0")/.."
wl=kitti
if [ "$1" = "kitti" ] || [ "$1" = "matterport" ]; then wl=$1; shift; fi
rounds=3
case "$1" in rounds=*) rounds=${1#rounds=}; shift;; esac
out=gpurun_out/ab
mkdir -p $out
: > $out/step_$wl.txt
for round in $(seq $rounds); do
  for spec in "$@"; do
    name=${spec%%@*}
    envs=""
    [ "$spec" != "$name" ] && envs=$(echo "${spec#*@}" | tr ',' ' ')
    lib="$PWD/footprints_amd/libfootprints_hip.so"
    [ "$name" != "default" ] && lib="$PWD/scripts/ubench/bin/lib_$name.so"
    echo -n "$spec " >> $out/step_$wl.txt
    env $envs FP_LIB=$lib timeout 300 python bench.py --leg train-only --workload $wl --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['img_per_s'], d['ms_per_step'], d['final_loss'])" >> $out/step_$wl.txt 2>&1
  done
done
cat $out/step_$wl.txt
