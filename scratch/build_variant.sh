#!/bin/bash
# build a variant of libuc2_hip.so with extra -D flags for ONE source file: scratch/build_variant.sh <name> <file.hip> <flags...>
# -> ablibs/lib_<name>.so (all other objects from the regular build); select with UC2_LIB_PATH
set -e
cd "$(dirname "$0")/../uc2_amd/csrc"
name=$1; src=$2; shift 2
make -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c $src -o build/var_${name}.o
objs=$(ls build/*.o | grep -v "build/var_" | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ablibs/lib_${name}.so $objs build/var_${name}.o -ldl
rm -f build/var_${name}.o
echo built ablibs/lib_${name}.so
