# usage: bash tools/build_variant.sh <name> [-DFLAG ...]   -> ab_libs/<name>/{libmi355rt.so,libmi355rt_host.so}
# A build of the device library with extra preprocessor flags (applied to every file of it), for A/B
# timing (tools/ab_multi.sh).
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
mkdir -p ab_libs/$NAME build/obj/$NAME
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o build/obj/$NAME/bvh_build.o raytracer_amd/csrc/bvh_build.cpp &
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o build/obj/$NAME/bvh_device.o raytracer_amd/csrc/bvh_device.hip &
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o build/obj/$NAME/rt_api.o raytracer_amd/csrc/rt_api.hip &
wait
/opt/rocm/bin/hipcc $FLAGS -shared -o ab_libs/$NAME/libmi355rt.so build/obj/$NAME/rt_api.o build/obj/$NAME/bvh_build.o build/obj/$NAME/bvh_device.o
cp raytracer_amd/lib/libmi355rt_host.so ab_libs/$NAME/
