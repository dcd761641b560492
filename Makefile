# Build of the three in-tree shared objects (all git-ignored, all shipped to the GPU box by gpurun):
#   hydracore_amd/lib/libhydra_hip.so   the product: HIP kernels + C-ABI (include/hydra_hip.h), gfx950 only
#   hydracore_amd/lib/libhydra_host.so  C++ host layer: IHWLayer-shaped adapter, scene front end, BVH4 builder
#   oracle/liboracle.so                 CPU oracle (test infrastructure, never linked by the two above)
HIPCC    ?= hipcc
CXX      ?= g++
CC       ?= gcc
ARCH     ?= gfx950
LIBDIR   ?= hydracore_amd/lib
# EXTRA_DEFS: compile-time tuning (-DHK_LDS_DEPTH=.. -DHK_TRACE_MIN_BLOCKS=..); LIBDIR can point at a variant directory
EXTRA_DEFS ?=
HIPFLAGS := --offload-arch=$(ARCH) -O3 -ffp-contract=off -std=c++17 -fPIC $(EXTRA_DEFS)
CXXFLAGS := -std=c++17 -O2 -Wall -Wextra -Wno-unused-parameter -fPIC
HOSTSRC  := $(wildcard hydracore_amd/host/*.cpp)
HOSTHDR  := $(wildcard hydracore_amd/host/*.h) $(wildcard include/*.h)
# one object per translation unit, so that `make -j` compiles the device code in parallel (hk_kernels.h says which kernels live where)
HIPSRC   := $(wildcard hydracore_amd/csrc/*.hip)
HIPHDR   := $(wildcard hydracore_amd/csrc/*.h) $(wildcard include/*.h)
OBJDIR   ?= build/obj$(subst /,_,$(LIBDIR))
HIPOBJ   := $(patsubst hydracore_amd/csrc/%.hip,$(OBJDIR)/%.o,$(HIPSRC))
MAKEFLAGS += -j8

all: $(LIBDIR)/libhydra_hip.so $(LIBDIR)/libhydra_host.so oracle/liboracle.so oracle/liboracle_fast.so

# the BVH builder and the image kernels read none of the shading headers: they rebuild only when their own sources do
$(OBJDIR)/hydra_bvh.o $(OBJDIR)/hydra_img.o: $(OBJDIR)/%.o: hydracore_amd/csrc/%.hip hydracore_amd/csrc/hk_common.h $(wildcard include/*.h)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
# procedural textures: the device headers the run-time compiled kernel is built against, embedded as one text (hydra_proctex.hip hands it to hiprtc).
# Local #include lines are dropped (the files follow each other in dependency order); hiprtc supplies the HIP runtime declarations itself.
AMALGAM_SRC := include/hydra_layouts.h hydracore_amd/csrc/hk_common.h hydracore_amd/csrc/hk_trace.h hydracore_amd/csrc/hk_shading.h hydracore_amd/csrc/hk_proctex_rt.h
$(OBJDIR)/hk_proctex_amalgam.inc: $(AMALGAM_SRC)
	@mkdir -p $(OBJDIR)
	( echo 'R"HKAMALGAM(' ; \
	  echo 'typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t; typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;' ; \
	  echo '#define FLT_MAX 3.402823466e+38F' ; \
	  sed -e '/^[[:space:]]*#[[:space:]]*include/d' -e '/^#pragma once/d' $(AMALGAM_SRC) ; \
	  echo ')HKAMALGAM"' ) > $@
$(OBJDIR)/hydra_proctex.o: hydracore_amd/csrc/hydra_proctex.hip $(OBJDIR)/hk_proctex_amalgam.inc $(HIPHDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -I$(OBJDIR) -c $< -o $@
$(OBJDIR)/%.o: hydracore_amd/csrc/%.hip $(HIPHDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libhydra_hip.so: $(HIPOBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared $(HIPOBJ) -o $@ -lhiprtc

$(LIBDIR)/libhydra_host.so: $(HOSTSRC) $(HOSTHDR) $(LIBDIR)/libhydra_hip.so
	$(CXX) $(CXXFLAGS) -shared $(HOSTSRC) -o $@ -L$(LIBDIR) -lhydra_hip -Wl,-rpath,'$$ORIGIN'

oracle/liboracle.so: oracle/hydra_oracle.c oracle/hydra_oracle.h
	$(CC) -std=gnu11 -O2 -Wall -Wextra -fopenmp -ffp-contract=off -fPIC -shared oracle/hydra_oracle.c -o $@ -lm

# the same restatement as bench.py's CPU baseline times it: -O3 and the traversal's visit counters compiled out (never used as the checker)
oracle/liboracle_fast.so: oracle/hydra_oracle.c oracle/hydra_oracle.h
	$(CC) -std=gnu11 -O3 -DORC_NO_STATS -Wall -Wextra -fopenmp -ffp-contract=off -fPIC -shared oracle/hydra_oracle.c -o $@ -lm

clean:
	rm -f $(LIBDIR)/*.so oracle/liboracle.so oracle/liboracle_fast.so
	rm -rf build/obj*

.PHONY: all clean
