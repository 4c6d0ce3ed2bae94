#!/bin/bash
# build a variant of the library with extra flags for ONE source: tools/dbg/build_variant.sh NAME SOURCE.hip "-DFLAG=1 ..."
# -> tools/dbg/libquip_NAME.so (load it with QUIP_LIB_PATH); the other objects come from quip_for_all_amd/lib/obj
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; flags=$3
obj=/tmp/variant_${name}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c quip_for_all_amd/csrc/$src -o $obj
others=$(ls quip_for_all_amd/lib/obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libquip_${name}.so $obj $others
echo built tools/dbg/libquip_${name}.so
