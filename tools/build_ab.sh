#!/bin/bash
# tools/build_ab.sh <name> <file.hip> [-Dmacro=value ...]: a copy of the library with ONE translation unit rebuilt under other macros -> tmp_ab/lib<name>.so
# (for tools/flow_lib_ab.sh and friends: several builds timed on one box; the other objects are those of the last `make`)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); name=$1; src=$2; shift 2
mkdir -p $R/tmp_ab; cd $R/vpp_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Wno-cuda-compat"
[ "$src" = pyrlk.hip ] && F="$F -fno-slp-vectorize"
hipcc $F "$@" -c $src -o /tmp/ab_${name}_${src%.hip}.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tmp_ab/lib$name.so $(ls *.o | grep -v "^${src%.hip}.o$") /tmp/ab_${name}_${src%.hip}.o
echo "tmp_ab/lib$name.so"
