#!/bin/bash
# A/B builds of one translation unit of libmapdit_hip.so with an extra -D flag, linked against the objects of the current build:
#   tools/build_probe.sh attention SB_PROBE 1 2 3      ->  tools/_ab/lib_SB_PROBE1.so ...   (run with MAPDIT_LIB=<that file>)
set -e
cd "$(dirname "$0")/../map-dit_amd/csrc"
tu=$1; flag=$2; shift 2
mkdir -p ../../tools/_ab /tmp/mapdit_ab
objs=$(ls *.o | grep -v "^$tu.o$" | tr '\n' ' ')
for v in "$@"; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -fno-vectorize \
      -D$flag=$v -c $tu.hip -o /tmp/mapdit_ab/${tu}_$flag$v.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/lib_$flag$v.so $objs /tmp/mapdit_ab/${tu}_$flag$v.o -ldl ) &
done
wait
ls -la ../../tools/_ab/
