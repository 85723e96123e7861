#!/bin/bash
# build a variant of the library with extra -D flags on render.hip: tools/build_variant.sh <name> <flags...>
set -e
name=$1; shift
cd "$(dirname "$0")/../fredholm_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" -c render.hip -o /tmp/render_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libfredholm_hip_$name.so capi.o /tmp/render_$name.o bvh_build.o kat.o post.o
ls -la ../libfredholm_hip_$name.so
