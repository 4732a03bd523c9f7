#!/bin/bash
# build_variant_of.sh SOURCE NAME [-Dmacro=value ...]: liblsk with csrc/SOURCE.hip compiled under extra
# macros, into tools/exp/liblsk_NAME.so (run experiments on it with LSK_LIBRARY=...).
set -e
src=$1; name=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
obj=/tmp/${src}_$name.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops \
  -I $root/include -I $root/legionsolvers_amd/csrc "$@" -c $root/legionsolvers_amd/csrc/$src.hip -o $obj 2>&1 | grep -v packed-fp32 || true
objs=$(ls $root/legionsolvers_amd/build/*.o | grep -v "/$src.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs $obj -o $root/tools/exp/liblsk_$name.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built $root/tools/exp/liblsk_$name.so
