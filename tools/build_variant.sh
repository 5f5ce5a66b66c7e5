# usage: bash tools/build_variant.sh <name> [-DFLAG ...]   -> ab_libs/<name>/{libmi355rt.so,libmi355rt_host.so}
# A build of the device library with extra preprocessor flags, for A/B timing (tools/ab_libs.sh).
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
mkdir -p ab_libs/$NAME build/obj
[ build/obj/bvh_build.o -nt raytracer_amd/csrc/bvh_build.cpp ] || /opt/rocm/bin/hipcc $FLAGS -c -o build/obj/bvh_build.o raytracer_amd/csrc/bvh_build.cpp
[ build/obj/bvh_device.o -nt raytracer_amd/csrc/bvh_device.hip ] || /opt/rocm/bin/hipcc $FLAGS -c -o build/obj/bvh_device.o raytracer_amd/csrc/bvh_device.hip
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o build/obj/rt_api_$NAME.o raytracer_amd/csrc/rt_api.hip
/opt/rocm/bin/hipcc $FLAGS -shared -o ab_libs/$NAME/libmi355rt.so build/obj/rt_api_$NAME.o build/obj/bvh_build.o build/obj/bvh_device.o
cp raytracer_amd/lib/libmi355rt_host.so ab_libs/$NAME/
