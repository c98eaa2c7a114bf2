#!/bin/bash
# An alternative build of ONE translation unit with extra -D flags, linked with the product objects into tools/dbg/<name>/libaudiodec_hip.so
# (not tracked; travels to the GPU box with the snapshot).  A/B in one process environment: ADK_LIB_PATH=<that .so>.
#   tools/alt_build.sh <name> <unit without .hip> <flags...>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; unit=$2; shift 2
out=$R/tools/dbg/$name
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include "$@" -c $R/audiodec_amd/csrc/$unit.hip -o $out/$unit.o
objs=""
for s in $R/audiodec_amd/csrc/*.hip; do b=$(basename $s .hip); if [ "$b" = "$unit" ]; then objs="$objs $out/$unit.o"; else objs="$objs $R/audiodec_amd/csrc/.obj/$b.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $out/libaudiodec_hip.so $objs
echo $out/libaudiodec_hip.so
