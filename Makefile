# Builds the product library (HIP, gfx950 only) and the test oracle (plain C).
#   make            -> oat_amd/lib/liboatgpu.so + oracle/liboat_oracle.so
#   make host       -> C++ drop-in binaries under build/bin
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off is a PARITY requirement (the reference's CPU build rounds every
# mul/add of the MOG2 update separately), not a tuning choice.
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result -Wno-unused-value
CSRC     = oat_amd/csrc
LIB      = oat_amd/lib/liboatgpu.so
OBJS     = $(CSRC)/kernels_mog.o $(CSRC)/kernels_blob.o $(CSRC)/kernels_kalman.o $(CSRC)/oatgpu_api.o

all: $(LIB) oracle

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/oatgpu_internal.h include/oatgpu.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p oat_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)
	python3 tools/isa_hazard_check.py $@      # the gfx950 wide-store data hazard: the binary is checked, DESIGN.md 3b

oracle:
	$(MAKE) -C oracle

host: $(LIB) tools
	$(MAKE) -C oat_amd/host

# measurement aids: the pipelined loop from a plain C++ process, and a launch/event cost microbenchmark
tools: build/bin/bench_native build/bin/launch_gap
build/bin/bench_native: tools/bench_native.hip include/oatgpu.h $(LIB)
	@mkdir -p build/bin
	$(HIPCC) --offload-arch=$(ARCH) -O2 -w $< -Iinclude -Loat_amd/lib -loatgpu -Wl,-rpath,'$$ORIGIN/../../oat_amd/lib' -o $@
build/bin/launch_gap: tools/launch_gap.hip
	@mkdir -p build/bin
	$(HIPCC) --offload-arch=$(ARCH) -O2 -w $< -o $@

clean:
	rm -f $(OBJS) $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle host tools clean

# A/B variant builds: make variant NAME=px2 DEFS="-DOATGPU_PX=2"  -> build/variants/liboatgpu_px2.so
# Only these builds (-DOATGPU_MEASURE) read the OATGPU_EXPT / SERIAL / NB / ... measurement switches from the
# environment; the product library ignores them.  They live under build/ (git-ignored, but they travel to the GPU box):
# oat_amd/lib/ holds the product library and nothing else, and the Python binding loads another library only with
# OATGPU_MEASURE_PY=1 OATGPU_LIB=<path> (oat_amd/ffi.py; the tools/*.sh A/B scripts set both).
variant:
	@mkdir -p build/$(NAME) build/variants
	for f in kernels_mog kernels_blob kernels_kalman oatgpu_api; do $(HIPCC) $(HIPFLAGS) -DOATGPU_MEASURE $(DEFS) -c $(CSRC)/$$f.hip -o build/$(NAME)/$$f.o || exit 1; done
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o build/variants/liboatgpu_$(NAME).so build/$(NAME)/*.o
clean-variants:
	rm -rf build/variants oat_amd/lib/liboatgpu_*.so
.PHONY: variant clean-variants
