# Builds every native artefact in-tree:
#   raytracer_amd/lib/libmi355rt_host.so  host front-end (g++)
#   raytracer_amd/lib/libmi355rt.so       BVH build + HIP render path for gfx950 (hipcc)
#   raytracer_amd/lib/raytracer           command-line driver (links both)
#   oracle/liboracle.so                   CPU restatement of the reference (test infrastructure)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950
LIBDIR := raytracer_amd/lib

HOST_SRC := raytracer_amd/host/ron.cpp raytracer_amd/host/loader.cpp raytracer_amd/host/hostlib.cpp \
            raytracer_amd/host/jpeg.cpp
HOST_HDR := raytracer_amd/host/ron.h raytracer_amd/host/host_scene.h raytracer_amd/host/jpeg.h raytracer_amd/csrc/philox.h \
            raytracer_amd/csrc/scene_validate.h include/mi355rt.h
DEV_SRC := raytracer_amd/csrc/rt_api.hip raytracer_amd/csrc/bvh_build.cpp raytracer_amd/csrc/bvh_device.hip
DEV_HDR := $(wildcard raytracer_amd/csrc/*.h) include/mi355rt.h

# -ffp-contract=off everywhere: the reference is Rust, which never fuses a*b+c.
CXXFLAGS := -O2 -std=c++17 -fPIC -Wall -Wextra -ffp-contract=off -fno-fast-math
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math \
            -Wall -Wno-unused-function

all: host device cli oracle

host: $(LIBDIR)/libmi355rt_host.so
device: $(LIBDIR)/libmi355rt.so
cli: $(LIBDIR)/raytracer
oracle:
	$(MAKE) -C oracle

$(LIBDIR)/libmi355rt_host.so: $(HOST_SRC) $(HOST_HDR)
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -lz

# one object per source, so that `make -j` compiles them side by side and an edit rebuilds one
DEV_OBJ := $(patsubst raytracer_amd/csrc/%,build/obj/%.o,$(DEV_SRC))
build/obj/%.o: raytracer_amd/csrc/% $(DEV_HDR)
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
$(LIBDIR)/libmi355rt.so: $(DEV_OBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(DEV_OBJ)

$(LIBDIR)/raytracer: raytracer_amd/host/main.cpp $(LIBDIR)/libmi355rt_host.so $(LIBDIR)/libmi355rt.so
	$(CXX) $(CXXFLAGS) -o $@ raytracer_amd/host/main.cpp -L$(LIBDIR) -lmi355rt_host -lmi355rt \
	    -Wl,-rpath,'$$ORIGIN'

# Device listing of the render path and the check that no vector-register spill sits ahead of
# the EXEC restore of its block (tools/check_spills.py says what that breaks).
$(LIBDIR)/rt_api.s: $(DEV_SRC) $(DEV_HDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o $@ raytracer_amd/csrc/rt_api.hip
spillcheck: $(LIBDIR)/rt_api.s
	python3 tools/check_spills.py $(LIBDIR)/rt_api.s

clean:
	rm -rf $(LIBDIR) oracle/liboracle.so oracle/_ref

.PHONY: all host device cli oracle clean spillcheck

# development build with loop-trip counters (tools/lane_profile.py); not shipped
laneprof:
	@mkdir -p build/laneprof
	cp $(LIBDIR)/libmi355rt_host.so build/laneprof/
	$(HIPCC) $(HIPFLAGS) -DRT_LANE_PROFILE -shared -o build/laneprof/libmi355rt.so $(DEV_SRC)
