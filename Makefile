# Builds the C-ABI HIP library for gfx950 (MI355X) in-tree, and the CPU oracle has no native part.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := vqa_amd/csrc
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(SRCS:.hip=.o)
LIB   := vqa_amd/libvqa_hip.so
# -packed-fp32-ops: v_pk_{add,mul,fma}_f32 scalarised.  Beside MFMAs of the waves sharing the SIMD a packed f32 op costs
# far more than the two plain ops it replaces (same-box A/B of the whole step: 4.555 -> 4.531 ms, data gradients -2 %).
FLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops

all: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h
	$(HIPCC) $(FLAGS) -c $< -o $@

# one typed thunk per stream-taking entry point, generated from the header's declarations
$(CSRC)/plan_thunks.inc: include/vqa_hip.h scripts/gen_plan_thunks.py
	python3 scripts/gen_plan_thunks.py

$(CSRC)/plan.o: $(CSRC)/plan_thunks.inc include/vqa_hip.h

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

clean:
	rm -f $(OBJS) $(LIB)

.PHONY: all clean
