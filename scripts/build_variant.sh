#!/bin/bash
# Build an A/B variant of the library: scripts/build_variant.sh <name> "<a.hip b.hip ...>" [extra hipcc flags...]
# -> scripts/ubench/bin/lib_<name>.so (same ABI; select with FP_LIB=...).  The regular objects must be built already.
set -e
name=$1; srcs=$2; shift 2
cd "$(dirname "$0")/../footprints_amd/csrc"
tmp=/tmp/fp_variant_$name; mkdir -p $tmp
excl=""
for src in $srcs; do
  flags=""
  [ "$src" = "data_path.hip" ] && flags="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-vectorize -I../../include -I. -Wno-unused-function $flags "$@" -c $src -o $tmp/${src%.hip}.o &
  excl="$excl -e ^${src%.hip}.o\$"
done
wait
objs=$(ls *.o | grep -v $excl)
mkdir -p ../../scripts/ubench/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $tmp/*.o -o ../../scripts/ubench/bin/lib_$name.so
echo built scripts/ubench/bin/lib_$name.so
