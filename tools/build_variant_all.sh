#!/bin/bash
# a second library with extra -D flags on EVERY device source (tools/build_variant.sh recompiles render.hip only): tools/build_variant_all.sh <name> <flags...>
# -> fredholm_amd/libfredholm_hip_<name>.so (git-ignored, travels to the GPU box; FH_LIB=... selects it)
set -e
name=$1; shift
cd "$(dirname "$0")/../fredholm_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Wno-unused-result"
mkdir -p /tmp/fhv_$name
[ -f gen/tables.inc ] || make gen/tables.inc
for f in capi render bvh_build kat post; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o /tmp/fhv_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libfredholm_hip_$name.so /tmp/fhv_$name/capi.o /tmp/fhv_$name/render.o /tmp/fhv_$name/bvh_build.o /tmp/fhv_$name/kat.o /tmp/fhv_$name/post.o
ls -la ../libfredholm_hip_$name.so
