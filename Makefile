# Builds librtcore.so (HIP kernels + C ABI, gfx950 only) and the CPU oracle.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := raytrace_amd/csrc
# -ffp-contract=off is part of the numeric contract (bit-reproducible hit records).
# -fno-slp-vectorize: the compiler's own pairing of scalar float operations into packed instructions costs more register
# moves than it saves in the triangle test and the ray set-up (measured: +2.6 %); the box test is packed by hand.
HIPFLAGS := -std=c++17 -O3 --offload-arch=$(ARCH) -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wextra -Wno-unused-parameter $(EXTRA_HIPFLAGS)
HOSTFLAGS := -std=c++17 -O2 -ffp-contract=off -fPIC -Wall -Wextra

LIB := raytrace_amd/librtcore.so

# Host side above the C ABI (scene loader, worker, CLI, HTTP server) and the small tools the tests drive.
HOST     := raytrace_amd/host
HOSTCXX  ?= g++
HOSTCXXFLAGS := -std=c++17 -O2 -ffp-contract=off -Wall -Wextra -Iinclude -I$(HOST)
HOST_OBJS := $(HOST)/obj_model.o $(HOST)/image_io.o $(HOST)/jpeg_decode.o $(HOST)/scene_loader.o $(HOST)/worker.o $(HOST)/http_server.o
BINS     := bin/main bin/server bin/obj_dump bin/scene_dump bin/png_tool bin/fuzz_readers
RPATH    := -Wl,-rpath,'$$ORIGIN/../raytrace_amd'

all: $(LIB) oracle host

host: $(BINS)

$(HOST)/%.o: $(HOST)/%.cpp $(wildcard $(HOST)/*.h) include/rtcore.h
	$(HOSTCXX) $(HOSTCXXFLAGS) -c -o $@ $<

bin/main: $(HOST)/main.cpp $(HOST_OBJS) $(LIB)
	@mkdir -p bin
	$(HOSTCXX) $(HOSTCXXFLAGS) -o $@ $(HOST)/main.cpp $(HOST_OBJS) -Lraytrace_amd -lrtcore -lz -lpthread $(RPATH)

bin/server: $(HOST)/server.cpp $(HOST_OBJS) $(LIB)
	@mkdir -p bin
	$(HOSTCXX) $(HOSTCXXFLAGS) -o $@ $(HOST)/server.cpp $(HOST_OBJS) -Lraytrace_amd -lrtcore -lz -lpthread $(RPATH)

bin/obj_dump: tools/host/obj_dump.cpp $(HOST)/obj_model.o
	@mkdir -p bin
	$(HOSTCXX) $(HOSTCXXFLAGS) -o $@ $^

bin/png_tool: tools/host/png_tool.cpp $(HOST)/image_io.o $(HOST)/jpeg_decode.o
	@mkdir -p bin
	$(HOSTCXX) $(HOSTCXXFLAGS) -o $@ $^ -lz

bin/scene_dump: tools/host/scene_dump.cpp $(HOST)/scene_loader.o $(HOST)/obj_model.o $(HOST)/image_io.o $(HOST)/jpeg_decode.o $(LIB)
	@mkdir -p bin
	$(HOSTCXX) $(HOSTCXXFLAGS) -o $@ tools/host/scene_dump.cpp $(HOST)/scene_loader.o $(HOST)/obj_model.o $(HOST)/image_io.o $(HOST)/jpeg_decode.o -Lraytrace_amd -lrtcore -lz $(RPATH)

# The readers of untrusted text (scene file, OBJ, MTL) once more, with AddressSanitizer and UBSan, under a mutation fuzzer
# (tests/test_host_fuzz.py runs it; ASAN_OPTIONS=detect_leaks=0 because the HIP runtime librtcore.so pulls in keeps its own).
bin/fuzz_readers: tools/host/fuzz_readers.cpp $(HOST)/obj_model.cpp $(HOST)/scene_loader.cpp $(HOST)/image_io.cpp $(HOST)/jpeg_decode.cpp $(wildcard $(HOST)/*.h) $(LIB)
	@mkdir -p bin
	$(HOSTCXX) -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off -Wall -Wextra -Iinclude -I$(HOST) -o $@ \
	    tools/host/fuzz_readers.cpp $(HOST)/obj_model.cpp $(HOST)/scene_loader.cpp $(HOST)/image_io.cpp $(HOST)/jpeg_decode.cpp -Lraytrace_amd -lrtcore -lz $(RPATH)

$(CSRC)/rt_kernels.o: $(CSRC)/rt_kernels.hip $(CSRC)/rt_build.h $(CSRC)/rt_device.h $(CSRC)/rt_host.h $(CSRC)/rt_packed.h include/rtcore.h
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(CSRC)/rt_build.o: $(CSRC)/rt_build.hip $(CSRC)/rt_build.h $(CSRC)/rt_host.h $(CSRC)/rt_packed.h include/rtcore.h
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(CSRC)/rt_multi.o: $(CSRC)/rt_multi.hip include/rtcore.h
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(CSRC)/rt_host.o: $(CSRC)/rt_host.cpp $(CSRC)/rt_host.h $(CSRC)/rt_packed.h include/rtcore.h
	$(HIPCC) $(HOSTFLAGS) -c -o $@ $<

$(LIB): $(CSRC)/rt_kernels.o $(CSRC)/rt_host.o $(CSRC)/rt_multi.o $(CSRC)/rt_build.o
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $^ -ldl

oracle:
	$(MAKE) -C oracle

resource-usage:
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c -o /dev/null $(CSRC)/rt_kernels.hip

clean:
	rm -f $(CSRC)/*.o $(HOST)/*.o $(LIB) $(BINS)
	$(MAKE) -C oracle clean

.PHONY: all oracle host clean resource-usage
