#!/bin/bash
# Build an A/B variant of the library: scripts/build_variant.sh <name> <file.hip> [extra hipcc flags...]
# -> scripts/ubench/bin/lib_<name>.so (same ABI; select with FP_LIB=...).  The regular objects must be built already.
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../footprints_amd/csrc"
tmp=/tmp/fp_variant_$name; mkdir -p $tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-vectorize -I../../include -I. -Wno-unused-function "$@" -c $src -o $tmp/${src%.hip}.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
mkdir -p ../../scripts/ubench/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $tmp/${src%.hip}.o -o ../../scripts/ubench/bin/lib_$name.so
echo built scripts/ubench/bin/lib_$name.so
