# Builds librtcore.so (HIP kernels + C ABI, gfx950 only) and the CPU oracle.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := raytrace_amd/csrc
# -ffp-contract=off is part of the numeric contract (bit-reproducible hit records).
HIPFLAGS := -std=c++17 -O3 --offload-arch=$(ARCH) -ffp-contract=off -fPIC -Wall -Wextra -Wno-unused-parameter
HOSTFLAGS := -std=c++17 -O2 -ffp-contract=off -fPIC -Wall -Wextra

LIB := raytrace_amd/librtcore.so

all: $(LIB) oracle

$(CSRC)/rt_kernels.o: $(CSRC)/rt_kernels.hip $(CSRC)/rt_host.h $(CSRC)/rt_packed.h include/rtcore.h
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(CSRC)/rt_host.o: $(CSRC)/rt_host.cpp $(CSRC)/rt_host.h $(CSRC)/rt_packed.h include/rtcore.h
	$(HIPCC) $(HOSTFLAGS) -c -o $@ $<

$(LIB): $(CSRC)/rt_kernels.o $(CSRC)/rt_host.o
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $^

oracle:
	$(MAKE) -C oracle

resource-usage:
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c -o /dev/null $(CSRC)/rt_kernels.hip

clean:
	rm -f $(CSRC)/*.o $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean resource-usage
