#!/bin/bash
# build_variant.sh <variant.hip> <name>: links tools/dbg/libv_<name>.so = the in-tree library with decode_block.o replaced by
# the variant source (A/B timing of the block engine through QUIP_LIB_PATH)
set -e
cd "$(dirname "$0")/../.."
O=quip_for_all_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iquip_for_all_amd/csrc -Iinclude -c "$1" -o /tmp/v_$2.o
objs=$(ls $O/*.o | grep -v decode_block.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libv_$2.so $objs /tmp/v_$2.o
echo built tools/dbg/libv_$2.so
