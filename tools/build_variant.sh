#!/bin/bash
# Kernel experiments: rebuild ONE translation unit with extra -D flags and link it with the objects of the regular build
# into cfdbench_amd/_C/libcfdbench_amd_<tag>.so; select it with CFDBENCH_AMD_LIB=<that path> (cfdbench_amd/_lib.py).
#   tools/build_variant.sh <tag> <file.hip> [-DFLAG ...]
set -e
tag=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
C=$root/cfdbench_amd/csrc; O=$root/cfdbench_amd/_C
python -c "from cfdbench_amd import build; build.build()" >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -I$C -I$root/include -Wno-unused-result "$@" -x hip -c $C/$src -o $O/$src.$tag.o 2> >(grep -v "warning\|^ *[0-9]* |\|\^\|generated" >&2)
objs=$(ls $O/*.hip.o $O/*.cpp.o | grep -v "/$src.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libcfdbench_amd_$tag.so $objs $O/$src.$tag.o
echo $O/libcfdbench_amd_$tag.so
