#!/bin/bash
# A/B of one engine switch on ONE box, alternating runs: scripts/ab_env.sh VAR A B [reps] [extra bench args]  (train-only leg, KITTI exact format)
VAR=$1; A=$2; B=$3; REPS=${4:-3}; shift 4
for i in $(seq $REPS); do
  for v in $A $B; do
    echo -n "$VAR=$v  "; env $VAR=$v python bench.py --leg train-only --steps 30 --warmup 8 "$@" 2>/dev/null | tail -1
  done
done
