# Builds everything that runs on the GPU box, in-tree:
#   raytracer_amd/lib/libhip_raytracer.so   the C ABI + gfx950 kernels   (make lib)
#   raytracer_amd/bin/raytracer_hip         host CLI over the C ABI       (make host)
#   oracle/_build, oracle/_ref              test-only checkers            (make oracle)
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off: fusion happens only where the source asks for it (exact vs fast paths)
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wextra -Iinclude
HOSTCXX  ?= /opt/rocm/lib/llvm/bin/clang++
# -ffp-contract=off: the software backend evaluates IEEE fp32 in source order, like the reference's build (build.sh:7-31)
HOSTFLAGS := -std=c++20 -O2 -ffp-contract=off -Wall -Wextra -pedantic -Werror -Iinclude -Iraytracer_amd/host

LIB := raytracer_amd/lib/libhip_raytracer.so
CLI := raytracer_amd/bin/raytracer_hip
HOST_SRCS := $(wildcard raytracer_amd/host/*.cpp)

.PHONY: all lib host oracle tools timeline clean
all: lib host oracle tools

lib: $(LIB)
$(LIB): raytracer_amd/csrc/hip_raytracer_kernels.hip raytracer_amd/csrc/rt_exact_math.h include/hip_raytracer.h
	@mkdir -p raytracer_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $<

# measurement build of the library: per-wave start / end stamps of the fast camera launch (tools/wave_timeline.py)
timeline: tools/bin/libhip_raytracer_timeline.so
tools/bin/libhip_raytracer_timeline.so: raytracer_amd/csrc/hip_raytracer_kernels.hip raytracer_amd/csrc/rt_exact_math.h include/hip_raytracer.h
	@mkdir -p tools/bin
	$(HIPCC) $(HIPFLAGS) -DAEHIP_WAVE_TIMELINE -shared -o $@ $<

# GPU-side self checks and microbenchmarks (run on the GPU box by tests / by hand)
tools: raytracer_amd/bin/check_exact_math tools/bin/valu_peak
raytracer_amd/bin/check_exact_math: tools/check_exact_math.hip raytracer_amd/csrc/rt_exact_math.h
	@mkdir -p raytracer_amd/bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -o $@ $<
tools/bin/valu_peak: tools/valu_peak.hip
	@mkdir -p tools/bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -o $@ $<

host: $(CLI)
$(CLI): $(HOST_SRCS) $(wildcard raytracer_amd/host/*.h) include/hip_raytracer.h $(LIB)
	@mkdir -p raytracer_amd/bin
	$(HOSTCXX) $(HOSTFLAGS) $(HOST_SRCS) -Lraytracer_amd/lib -lhip_raytracer -Wl,-rpath,'$$ORIGIN/../lib' -lpthread -o $@

oracle:
	$(MAKE) -C oracle all

clean:
	rm -rf raytracer_amd/lib raytracer_amd/bin
	$(MAKE) -C oracle clean
