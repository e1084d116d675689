# Top-level build: hipcc cross-compiles for gfx950 without a GPU.
#   make            -> icp_amd/libicp_amd.so, bin/GPUICP, oracle/liboracle.so (+ oracle/_ref when the reference is mounted)
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := icp_amd/csrc
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wextra -Wno-unused-parameter
LIB      := icp_amd/libicp_amd.so

all: $(LIB) bin/GPUICP oracle

$(CSRC)/kernels.o: $(CSRC)/kernels.hip $(CSRC)/kernels.h $(CSRC)/reduce.h $(CSRC)/exact.h $(CSRC)/search_common.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/search_mfma.o: $(CSRC)/search_mfma.hip $(CSRC)/kernels.h $(CSRC)/exact.h $(CSRC)/search_common.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/grid.o: $(CSRC)/grid.hip $(CSRC)/kernels.h $(CSRC)/reduce.h $(CSRC)/exact.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/devmem.o: $(CSRC)/devmem.cpp $(CSRC)/kernels.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/engine.o: $(CSRC)/engine.hip $(CSRC)/kernels.h include/icp_amd.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/hostio.o: $(CSRC)/hostio.cpp include/icp_amd.h
	$(CXX) -O2 -std=c++17 -fPIC -ffp-contract=off -Wall -Wextra -c $< -o $@

$(LIB): $(CSRC)/kernels.o $(CSRC)/search_mfma.o $(CSRC)/grid.o $(CSRC)/devmem.o $(CSRC)/engine.o $(CSRC)/hostio.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $^ -ldl

bin/GPUICP: $(CSRC)/gpuicp_main.cpp $(LIB) include/icp_amd.h
	@mkdir -p bin
	$(CXX) -O2 -std=c++17 -Wall -Wextra -pthread -o $@ $(CSRC)/gpuicp_main.cpp -Licp_amd -licp_amd -Wl,-rpath,'$$ORIGIN/../icp_amd'

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f $(CSRC)/*.o $(LIB) bin/GPUICP
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
