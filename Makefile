# Builds libsoapdenovo2_amd.so (C ABI of include/soapdenovo2_amd.h) and the two pregraph executables for gfx950.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := soapdenovo2_amd/csrc
OUT     := soapdenovo2_amd
CXX     ?= g++
CXXFLAGS := -O3 -std=c++17 -fPIC -Wall -Wno-unknown-pragmas -Wno-unused-function -Wno-unused-result -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value
# `make MEASURE=1` builds libsoapdenovo2_amd_measure.so beside the product: the same sources with -DPG_MEASURE, i.e. with the kernels' geometry
# knobs read from the environment (csrc/env.hpp: env_measure) and K2's phase-timer instantiation.  The product library has neither.
ifeq ($(MEASURE),1)
O       := m.o
LIBNAME := libsoapdenovo2_amd_measure.so
CXXFLAGS += -DPG_MEASURE
HIPFLAGS += -DPG_MEASURE
ifneq ($(K2_LOOK),)
HIPFLAGS += -DPG_K2_LOOK=$(K2_LOOK)
endif
ifneq ($(XDEF),)
HIPFLAGS += -D$(XDEF)
endif
else
O       := o
LIBNAME := libsoapdenovo2_amd.so
endif
HOSTOBJ := $(CSRC)/host_graph.$(O) $(CSRC)/host_reads.$(O) $(CSRC)/call_pregraph.$(O) $(CSRC)/host_skm.$(O) $(CSRC)/host_emu.$(O) $(CSRC)/host_plan.$(O) $(CSRC)/arena.$(O)
DEVOBJ  := $(CSRC)/pregraph_kernels.$(O) $(CSRC)/partition_kernels.$(O) $(CSRC)/graph_kernels.$(O) $(CSRC)/sort_records.$(O) $(CSRC)/exchange.$(O)
HDRS    := $(wildcard $(CSRC)/*.hpp) include/soapdenovo2_amd.h

ifeq ($(MEASURE),1)
all: $(OUT)/$(LIBNAME)
else
all: $(OUT)/libsoapdenovo2_amd.so $(OUT)/bin/SOAPdenovo-63mer $(OUT)/bin/SOAPdenovo-127mer $(OUT)/bin/synth_fastq
endif

$(CSRC)/%.$(O): $(CSRC)/%.cpp $(HDRS)
	$(CXX) $(CXXFLAGS) -c $< -o $@
$(CSRC)/%.$(O): $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT)/$(LIBNAME): $(HOSTOBJ) $(DEVOBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $^ -lz -lpthread -ldl

$(OUT)/bin/SOAPdenovo-63mer: $(CSRC)/main.cpp $(OUT)/libsoapdenovo2_amd.so
	@mkdir -p $(OUT)/bin
	$(CXX) $(CXXFLAGS) $< -o $@ -L$(OUT) -lsoapdenovo2_amd -Wl,-rpath,'$$ORIGIN/..'
$(OUT)/bin/SOAPdenovo-127mer: $(CSRC)/main.cpp $(OUT)/libsoapdenovo2_amd.so
	@mkdir -p $(OUT)/bin
	$(CXX) $(CXXFLAGS) -DPG_MER127 $< -o $@ -L$(OUT) -lsoapdenovo2_amd -Wl,-rpath,'$$ORIGIN/..'

# the deterministic FASTQ generator of the big whole-command checks (scripts/big_cli_check.py); not part of the library
$(OUT)/bin/synth_fastq: scripts/synth_fastq.cpp
	@mkdir -p $(OUT)/bin
	$(CXX) -O3 -std=c++17 -pthread $< -o $@

clean:
	rm -f $(CSRC)/*.o $(OUT)/libsoapdenovo2_amd.so $(OUT)/libsoapdenovo2_amd_measure.so $(OUT)/bin/SOAPdenovo-*
.PHONY: all clean
