# Builds libdlsa_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
HIPCC ?= hipcc
ARCH  ?= gfx950
CSRC  := dlsa_amd/csrc
OUT   := dlsa_amd/libdlsa_hip.so
SRCS  := $(CSRC)/error.cpp $(CSRC)/gram.hip $(CSRC)/gram_wide.hip $(CSRC)/gram_narrow.hip $(CSRC)/logit.hip $(CSRC)/dense.hip $(CSRC)/chol.hip $(CSRC)/synth.hip $(CSRC)/design.hip $(CSRC)/onehot.hip \
         $(CSRC)/irls.hip $(CSRC)/lars.hip
OBJS  := $(patsubst $(CSRC)/%,build/%.o,$(SRCS))
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Wall -Wno-unused-function

all: $(OUT)

build/%.o: $(CSRC)/% $(CSRC)/common.h include/dlsa_hip.h
	@mkdir -p build
	$(HIPCC) $(FLAGS) -x hip -c $< -o $@

$(OUT): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

clean:
	rm -rf build $(OUT)
.PHONY: all clean
