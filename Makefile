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
HIPFLAGS := --offload-arch=$(ARCH) -O3 -ffp-contract=off -std=c++17 -fPIC -shared $(EXTRA_DEFS)
CXXFLAGS := -std=c++17 -O2 -Wall -Wextra -Wno-unused-parameter -fPIC
HOSTSRC  := $(wildcard hydracore_amd/host/*.cpp)
HOSTHDR  := $(wildcard hydracore_amd/host/*.h) $(wildcard include/*.h)
HIPSRC   := hydracore_amd/csrc/hydra_hip.hip hydracore_amd/csrc/hydra_bvh.hip hydracore_amd/csrc/hydra_img.hip
HIPHDR   := $(wildcard hydracore_amd/csrc/*.h) $(wildcard include/*.h)

all: $(LIBDIR)/libhydra_hip.so $(LIBDIR)/libhydra_host.so oracle/liboracle.so

$(LIBDIR)/libhydra_hip.so: $(HIPSRC) $(HIPHDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) $(HIPSRC) -o $@

$(LIBDIR)/libhydra_host.so: $(HOSTSRC) $(HOSTHDR) $(LIBDIR)/libhydra_hip.so
	$(CXX) $(CXXFLAGS) -shared $(HOSTSRC) -o $@ -L$(LIBDIR) -lhydra_hip -Wl,-rpath,'$$ORIGIN'

oracle/liboracle.so: oracle/hydra_oracle.c oracle/hydra_oracle.h
	$(CC) -std=gnu11 -O2 -Wall -Wextra -fopenmp -ffp-contract=off -fPIC -shared oracle/hydra_oracle.c -o $@ -lm

clean:
	rm -f $(LIBDIR)/*.so oracle/liboracle.so

.PHONY: all clean
