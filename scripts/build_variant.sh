#!/bin/bash
# usage: [EXTRA_SRC=dev/msda_patch.hip] scripts/build_variant.sh <name> [-DZIRA_...=v ...]  -> build_ab/<name>.so (developer A/B builds; run here, they
# travel with gpurun; select with ZIRA_MSDA_LIB=build_ab/<name>.so).  The source list is the library's own (ziragroundingdino_amd/build.py).
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build_ab
src=$root/ziragroundingdino_amd/csrc
sources=$(cd $root && python3 -c "from ziragroundingdino_amd import build; print(' '.join(build.SOURCES))")
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -munsafe-fp-atomics -I$root/include "$@" $sources ${EXTRA_SRC:+$src/$EXTRA_SRC} -o $root/build_ab/$name.so
