# Builds libdlsa_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
#   make            the product library (no wrong-result timing knobs: common.h DLSA_DBG_WRONG is the constant 0)
#   make knobs      bench/libdlsa_hip_knobs.so with -DDLSA_DEBUG_KNOBS for timing experiments (never shipped / loaded by dlsa_amd)
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := dlsa_amd/csrc
OUT   ?= dlsa_amd/libdlsa_hip.so
BUILD ?= build
EXTRA ?=
SRCS  := $(CSRC)/error.cpp $(CSRC)/options.cpp $(CSRC)/comm.cpp $(CSRC)/gram.hip $(CSRC)/gram_wide.hip $(CSRC)/gram_narrow.hip $(CSRC)/gram_cyclic.hip $(CSRC)/gram_plan.hip $(CSRC)/logit.hip $(CSRC)/dense.hip $(CSRC)/chol.hip $(CSRC)/eigsolve.hip $(CSRC)/synth.hip $(CSRC)/design.hip $(CSRC)/onehot.hip \
         $(CSRC)/irls.hip $(CSRC)/irls_small.hip $(CSRC)/irls_pass.hip $(CSRC)/irls_batch.hip $(CSRC)/irls_wide.hip $(CSRC)/lars.hip $(CSRC)/lars_q.hip $(CSRC)/lars_c.hip
# the plan-driven fp64 Gram kernel: one translation unit per range of widths (C_LO_HI = CUs per slab group, full tiles),
# each compiled from gram_plan_unit.hip with the plans tools/gen_gram_plan_asm.py writes into $(BUILD)/gen at build time
PLAN_UNITS := 1_8_17 2_18_24 4_25_28 4_29_32 4_33_35
GEN   := $(BUILD)/gen
OBJS  := $(patsubst $(CSRC)/%,$(BUILD)/%.o,$(SRCS)) $(patsubst %,$(BUILD)/gram_plan_unit_%.o,$(PLAN_UNITS)) $(BUILD)/lars_t512.o
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Wall -Wno-unused-function $(EXTRA)
# -Wno-inline-asm ONLY for the translation units whose generated MFMA blocks name AGPRs beyond a127 in kernels bounded to
# two waves per SIMD: hipcc calls them "reserved" but allocates them.  What the warning would have guarded is checked after
# the build instead, on the shipped code objects (`make check` = tools/check_agpr_kernels.py: no scratch, planned AGPR
# counts, VGPR + AGPR <= 256, no accumulator moves inside a loop).
AGPR_UNITS := gram_narrow.hip gram_cyclic.hip irls_pass.hip
$(patsubst %,$(BUILD)/%.o,$(AGPR_UNITS)) $(patsubst %,$(BUILD)/gram_plan_unit_%.o,$(PLAN_UNITS)): FLAGS += -Wno-inline-asm

all: $(OUT)

$(BUILD)/%.o: $(CSRC)/% $(CSRC)/common.h $(wildcard $(CSRC)/*.inc) $(wildcard $(CSRC)/*.h) include/dlsa_hip.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(FLAGS) -x hip -c $< -o $@

# the LARS path kernels a second time with 512-thread workgroups (lars.hip: the C ABI entry picks the build by p)
$(BUILD)/lars_t512.o: $(CSRC)/lars.hip $(CSRC)/common.h include/dlsa_hip.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(FLAGS) -DDLSA_LARS_THREADS=512 -DDLSA_LARS_SECONDARY -x hip -c $< -o $@

$(GEN)/gram_plan_common.inc: tools/gen_gram_plan_asm.py
	@mkdir -p $(GEN)
	python3 $< common > $@

$(GEN)/gram_plan_%.inc: tools/gen_gram_plan_asm.py
	@mkdir -p $(GEN)
	python3 $< plans $(subst _, ,$*) > $@

.SECONDARY: $(patsubst %,$(GEN)/gram_plan_%.inc,$(PLAN_UNITS))
$(BUILD)/gram_plan_unit_%.o: $(CSRC)/gram_plan_unit.hip $(GEN)/gram_plan_%.inc $(GEN)/gram_plan_common.inc $(CSRC)/gram_plan_kernel.inc $(CSRC)/gram_plan.h $(CSRC)/common.h
	$(HIPCC) $(FLAGS) -I$(GEN) -DPLAN_LO=$(word 2,$(subst _, ,$*)) -DPLAN_HI=$(word 3,$(subst _, ,$*)) -DPLAN_INC='"gram_plan_$*.inc"' -x hip -c $< -o $@

$(OUT): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--no-undefined $(OBJS) -ldl -o $@

check: $(OUT)
	python3 tools/check_agpr_kernels.py $(OUT)

knobs:
	$(MAKE) BUILD=build/knobs OUT=bench/libdlsa_hip_knobs.so EXTRA=-DDLSA_DEBUG_KNOBS

clean:
	rm -rf build $(OUT) bench/libdlsa_hip_knobs.so
.PHONY: all clean knobs check
