# Host code (RON parser, loader, JPEG decoder, PNG writer, flatten + BVH build) under ASan + UBSan.
# usage: bash tools/host_sanitize.sh <directory with .ron / .jpg files>
set -e
cd "$(dirname "$0")/.."
mkdir -p build/san
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -ffp-contract=off -pthread -Iinclude -o build/san/host_sanitize tools/host_sanitize.cpp \
    raytracer_amd/host/ron.cpp raytracer_amd/host/loader.cpp raytracer_amd/host/hostlib.cpp \
    raytracer_amd/host/jpeg.cpp raytracer_amd/csrc/bvh_build.cpp -lz
ASAN_OPTIONS=detect_leaks=1 ./build/san/host_sanitize "$1" build/san
