# Builds libsoapdenovo2_amd.so (C ABI of include/soapdenovo2_amd.h) and the two pregraph executables for gfx950.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := soapdenovo2_amd/csrc
OUT     := soapdenovo2_amd
CXX     ?= g++
CXXFLAGS := -O3 -std=c++17 -fPIC -Wall -Wno-unknown-pragmas -Wno-unused-function -Wno-unused-result -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value
HOSTOBJ := $(CSRC)/host_graph.o $(CSRC)/host_reads.o $(CSRC)/call_pregraph.o $(CSRC)/host_skm.o $(CSRC)/host_emu.o
DEVOBJ  := $(CSRC)/pregraph_kernels.o $(CSRC)/partition_kernels.o $(CSRC)/graph_kernels.o $(CSRC)/sort_records.o $(CSRC)/exchange.o
HDRS    := $(wildcard $(CSRC)/*.hpp) include/soapdenovo2_amd.h

all: $(OUT)/libsoapdenovo2_amd.so $(OUT)/bin/SOAPdenovo-63mer $(OUT)/bin/SOAPdenovo-127mer $(OUT)/bin/synth_fastq

$(CSRC)/%.o: $(CSRC)/%.cpp $(HDRS)
	$(CXX) $(CXXFLAGS) -c $< -o $@
$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT)/libsoapdenovo2_amd.so: $(HOSTOBJ) $(DEVOBJ)
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
	rm -f $(CSRC)/*.o $(OUT)/libsoapdenovo2_amd.so $(OUT)/bin/SOAPdenovo-*
.PHONY: all clean
