# oracle/ref.mk — builds the UNMODIFIED reference (ggml @ 2025-02-13) from the sources where they
# lie under $(REF) into oracle/_ref/.  Test infrastructure only: the product never links these.
# No reference source is copied; this is a hand-written recipe (the reference's CMake is not run).
#
#   make -f oracle/ref.mk            (from the repo root)
#
# Outputs (git-ignored, but they travel to the GPU box with gpurun):
#   oracle/_ref/libggml-base.so  libggml-cpu.so  libggml.so
#   oracle/_ref/test-backend-ops  test-quantize-fns  test-mul-mat   (reference tests, unmodified)
#   oracle/_ref/cpu_baseline      (oracle/cpu_baseline.cpp: times the reference CPU backend on one MUL_MAT)
#   oracle/_ref/gpt-2-quantize    (examples/gpt-2/quantize.cpp, unmodified)
#   oracle/_ref/gpt2_harness      (oracle/gpt2_harness.cpp: includes examples/gpt-2/main-backend.cpp verbatim)
#   oracle/_ref/sched_harness     (oracle/sched_harness.cpp: includes examples/gpt-2/main-sched.cpp verbatim; ggml_backend_sched)
#   oracle/_ref/split_harness     (oracle/split_harness.cpp: split buffer type, async copies, events, pinned host buffers via the public API)
#
# ISA flags: x86-64-v3 (AVX2+FMA+F16C) instead of the reference's default -march=native so the same
# binaries run on this container's Xeon and on the GPU box's host CPU.  That selects the AVX2 bodies
# of quantize_row_q8_0 / ggml_vec_dot_* (src/ggml-cpu/ggml-cpu-quants.c), the ones the oracle in
# oracle/ggml_oracle.c restates.
REF   ?= /root/reference
OUT   ?= oracle/_ref
CC    ?= gcc
CXX   ?= g++
ARCH  ?= -march=x86-64-v3
OPT   ?= -O3 -DNDEBUG
DEFS  := -D_GNU_SOURCE -D_XOPEN_SOURCE=600 -DGGML_SCHED_MAX_COPIES=4 -DGGML_SHARED -DGGML_BACKEND_SHARED
INC   := -I$(REF)/include -I$(REF)/src -I$(REF)/src/ggml-cpu
CFLAGS_COMMON   := $(OPT) -fPIC -std=gnu11   $(DEFS) $(INC) -w
CXXFLAGS_COMMON := $(OPT) -fPIC -std=gnu++17 $(DEFS) $(INC) -w

BASE_C   := ggml.c ggml-alloc.c ggml-quants.c
BASE_CXX := ggml-backend.cpp ggml-opt.cpp ggml-threading.cpp gguf.cpp
CPU_C    := ggml-cpu/ggml-cpu.c ggml-cpu/ggml-cpu-quants.c
CPU_CXX  := ggml-cpu/ggml-cpu.cpp ggml-cpu/ggml-cpu-aarch64.cpp ggml-cpu/ggml-cpu-hbm.cpp \
            ggml-cpu/ggml-cpu-traits.cpp ggml-cpu/amx/amx.cpp ggml-cpu/amx/mmq.cpp

BASE_OBJ := $(patsubst %,$(OUT)/obj/base/%.o,$(BASE_C) $(BASE_CXX))
CPU_OBJ  := $(patsubst %,$(OUT)/obj/cpu/%.o,$(CPU_C) $(CPU_CXX))

LIBS  := $(OUT)/libggml-base.so $(OUT)/libggml-cpu.so $(OUT)/libggml.so
BINS  := $(OUT)/test-backend-ops $(OUT)/test-quantize-fns $(OUT)/test-mul-mat $(OUT)/cpu_baseline $(OUT)/gpt-2-quantize $(OUT)/gpt2_harness $(OUT)/sched_harness $(OUT)/split_harness $(OUT)/synth_data $(OUT)/v4/cpu_baseline

all: $(LIBS) $(BINS)

$(OUT)/obj/base/%.c.o: $(REF)/src/%.c
	@mkdir -p $(dir $@)
	$(CC) $(CFLAGS_COMMON) -DGGML_BUILD -c $< -o $@
$(OUT)/obj/base/%.cpp.o: $(REF)/src/%.cpp
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS_COMMON) -DGGML_BUILD -c $< -o $@
$(OUT)/obj/cpu/%.c.o: $(REF)/src/%.c
	@mkdir -p $(dir $@)
	$(CC) $(CFLAGS_COMMON) $(ARCH) -fopenmp -DGGML_BACKEND_BUILD -DGGML_USE_OPENMP -DGGML_USE_CPU_AARCH64 -c $< -o $@
$(OUT)/obj/cpu/%.cpp.o: $(REF)/src/%.cpp
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS_COMMON) $(ARCH) -fopenmp -DGGML_BACKEND_BUILD -DGGML_USE_OPENMP -DGGML_USE_CPU_AARCH64 -c $< -o $@

$(OUT)/libggml-base.so: $(BASE_OBJ)
	$(CXX) -shared -o $@ $^ -lm -lpthread
$(OUT)/libggml-cpu.so: $(CPU_OBJ) $(OUT)/libggml-base.so
	$(CXX) -shared -fopenmp -o $@ $(CPU_OBJ) -L$(OUT) -lggml-base -Wl,-rpath,'$$ORIGIN'
$(OUT)/libggml.so: $(REF)/src/ggml-backend-reg.cpp $(OUT)/libggml-cpu.so
	$(CXX) $(CXXFLAGS_COMMON) -DGGML_BUILD -DGGML_USE_CPU -shared -o $@ $< -L$(OUT) -lggml-cpu -lggml-base -ldl -Wl,-rpath,'$$ORIGIN'

$(OUT)/test-%: $(REF)/tests/test-%.cpp $(LIBS)
	$(CXX) $(CXXFLAGS_COMMON) -o $@ $< -L$(OUT) -lggml -lggml-cpu -lggml-base -lpthread -Wl,-rpath,'$$ORIGIN'

$(OUT)/cpu_baseline: oracle/cpu_baseline.cpp $(LIBS)
	$(CXX) $(CXXFLAGS_COMMON) -o $@ $< -L$(OUT) -lggml -lggml-cpu -lggml-base -lpthread -Wl,-rpath,'$$ORIGIN'

EXC := $(REF)/examples/common.cpp $(REF)/examples/common-ggml.cpp
$(OUT)/gpt-2-quantize: $(REF)/examples/gpt-2/quantize.cpp $(EXC) $(LIBS)
	$(CXX) $(CXXFLAGS_COMMON) -I$(REF)/examples -o $@ $< $(EXC) -L$(OUT) -lggml -lggml-cpu -lggml-base -lpthread -Wl,-rpath,'$$ORIGIN'
$(OUT)/gpt2_harness: oracle/gpt2_harness.cpp $(EXC) $(LIBS)
	$(CXX) $(CXXFLAGS_COMMON) -I$(REF) -I$(REF)/examples -o $@ $< $(EXC) -L$(OUT) -lggml -lggml-cpu -lggml-base -lpthread -ldl -Wl,-rpath,'$$ORIGIN'

# examples/gpt-2/main-sched.cpp, unmodified, with the plug-in in its (compile-time) GPU slot: see oracle/sched_harness.cpp
$(OUT)/sched_harness: oracle/sched_harness.cpp $(EXC) $(LIBS)
	$(CXX) $(CXXFLAGS_COMMON) -DGGML_USE_CUDA -I$(REF) -I$(REF)/examples -o $@ $< $(EXC) -L$(OUT) -lggml -lggml-cpu -lggml-base -lpthread -Wl,-rpath,'$$ORIGIN'
$(OUT)/split_harness: oracle/split_harness.cpp $(LIBS)
	$(CXX) $(CXXFLAGS_COMMON) -o $@ $< -L$(OUT) -lggml -lggml-cpu -lggml-base -lpthread -Wl,-rpath,'$$ORIGIN'

$(OUT)/synth_data: oracle/synth_data.cpp $(LIBS)
	$(CXX) $(CXXFLAGS_COMMON) -o $@ $< -L$(OUT) -lggml-base -lpthread -Wl,-rpath,'$$ORIGIN'

# A second build of the CPU backend for the timing baseline only: AVX-512 + VNNI (x86-64-v4), the widest ISA common to this
# container's Xeon and the GPU box's EPYC 9575F — what the reference's default -march=native would select there.  The oracle
# and every parity test keep the x86-64-v3 build above (that is the build oracle/ggml_oracle.c restates).
ARCH4 ?= -march=x86-64-v4 -mavx512vnni -mavx512vbmi
CPU4_OBJ := $(patsubst %,$(OUT)/v4/obj/%.o,$(CPU_C) $(CPU_CXX))
$(OUT)/v4/obj/%.c.o: $(REF)/src/%.c
	@mkdir -p $(dir $@)
	$(CC) $(CFLAGS_COMMON) $(ARCH4) -fopenmp -DGGML_BACKEND_BUILD -DGGML_USE_OPENMP -DGGML_USE_CPU_AARCH64 -c $< -o $@
$(OUT)/v4/obj/%.cpp.o: $(REF)/src/%.cpp
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS_COMMON) $(ARCH4) -fopenmp -DGGML_BACKEND_BUILD -DGGML_USE_OPENMP -DGGML_USE_CPU_AARCH64 -c $< -o $@
$(OUT)/v4/libggml-cpu.so: $(CPU4_OBJ) $(OUT)/libggml-base.so
	$(CXX) -shared -fopenmp -o $@ $(CPU4_OBJ) -L$(OUT) -lggml-base -Wl,-rpath,'$$ORIGIN/..'
$(OUT)/v4/cpu_baseline: oracle/cpu_baseline.cpp $(OUT)/v4/libggml-cpu.so $(LIBS)
	$(CXX) $(CXXFLAGS_COMMON) -o $@ $< -L$(OUT)/v4 -lggml-cpu -L$(OUT) -lggml-base -lpthread -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,'$$ORIGIN/..'

clean:
	rm -rf $(OUT)
.PHONY: all clean
