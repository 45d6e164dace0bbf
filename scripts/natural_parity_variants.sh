#!/bin/bash
# GPU box: the natural-statistics / wide-range parity case under operand-format variants (which part of the hp format costs what)
out=${1:-gpurun_out/natvar}; mkdir -p $out
run() { name=$1; shift; env "$@" FP_PARITY_DUMP=$out/$name timeout 300 python -m pytest tests/test_gpu_parity_fullsize.py -q -k natural 2>&1 | tail -1 | sed "s/^/$name: /"; }
run hp FP_DUMMY=1
run t14 FP_LIB=scripts/ubench/bin/lib_t14.so
run p4 FP_LIB=scripts/ubench/bin/lib_p4.so
run hp_wgrad_exact FP_HP_WGRAD=0
run hp_tile_exact FP_HP_TILE=0
run bf3 FP_HP=0
