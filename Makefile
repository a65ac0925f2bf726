# Builds the gfx950 C-ABI library in-tree (the .so travels to the GPU box with the snapshot).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
PKG := d-vae_amd
CSRC := $(PKG)/csrc
LIB := $(PKG)/lib/libdvae_hip.so
HDRS := $(wildcard $(CSRC)/*.hpp) include/dvae_hip.h

all: $(LIB)

$(LIB): $(CSRC)/dvae_hip.hip $(HDRS)
	@mkdir -p $(PKG)/lib
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=1000000 \
	    -o $@ $(CSRC)/dvae_hip.hip

clean:
	rm -f $(LIB)

.PHONY: all clean
