# Build of the MI355X (gfx950) quasi-MCP solver and its test infrastructure.
#   make            -> lib (HIP C-ABI library + C++ host mirror) and the oracle
#   make lib        -> genome-downsampler_amd/lib/libqmcp_hip.so, libqmcp_host.so
#   make oracle     -> oracle/libqmcp_oracle.so               (test infrastructure)
#   make harness    -> tests/cpp/coverage_harness             (CoverageTester restatement)
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
CC       ?= gcc
ARCH     ?= gfx950
PKG      := genome-downsampler_amd
LIBDIR   := $(PKG)/lib
CSRC     := $(PKG)/csrc
HOST     := $(PKG)/host

HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function
CXXFLAGS := -O2 -std=c++17 -fPIC -ffp-contract=off -Wall -Iinclude -I$(HOST)/include -DHIP_ENABLED
CFLAGS   := -O3 -std=c11 -fPIC -Wall

.PHONY: all lib oracle harness clean
all: lib oracle harness

lib: $(LIBDIR)/libqmcp_hip.so $(LIBDIR)/libqmcp_host.so

$(LIBDIR)/qmcp_kernels.o: $(CSRC)/qmcp_kernels.hip $(CSRC)/qmcp_kernels.h $(wildcard $(CSRC)/kernels/*.inc.hip)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/qmcp_api.o: $(CSRC)/qmcp_api.hip $(wildcard $(CSRC)/api/*.inc.hip) $(CSRC)/qmcp_kernels.h include/qmcp_hip.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libqmcp_hip.so: $(LIBDIR)/qmcp_kernels.o $(LIBDIR)/qmcp_api.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $^ -o $@

HOST_SRCS := $(HOST)/src/bam_api.cpp $(HOST)/src/reads_gen.cpp $(HOST)/src/quasi_mcp_hip_solver.cpp \
             $(HOST)/src/amplicon_set.cpp $(HOST)/src/bam_io.cpp \
             $(HOST)/src/host_c_api.cpp
$(LIBDIR)/libqmcp_host.so: $(HOST_SRCS) $(wildcard $(HOST)/include/*.hpp $(HOST)/include/*/*.hpp) \
                           $(LIBDIR)/libqmcp_hip.so include/qmcp_hip.h
	$(CXX) $(CXXFLAGS) -shared $(HOST_SRCS) -L$(LIBDIR) -lqmcp_hip -lz -lpthread -Wl,-rpath,'$$ORIGIN' -o $@

oracle: oracle/libqmcp_oracle.so
oracle/libqmcp_oracle.so: oracle/qmcp_oracle.c oracle/qmcp_oracle.h
	$(CC) $(CFLAGS) -shared $< -o $@

harness: tests/cpp/coverage_harness
tests/cpp/coverage_harness: tests/cpp/coverage_harness.cpp $(LIBDIR)/libqmcp_host.so oracle/libqmcp_oracle.so
	$(CXX) $(CXXFLAGS) -Ioracle $< -L$(LIBDIR) -lqmcp_host -lqmcp_hip -Loracle -lqmcp_oracle \
	    -Wl,-rpath,'$$ORIGIN/../../$(LIBDIR)' -Wl,-rpath,'$$ORIGIN/../../oracle' -o $@

clean:
	rm -rf $(LIBDIR) oracle/libqmcp_oracle.so tests/cpp/coverage_harness
