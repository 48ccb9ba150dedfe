# Builds the gfx950 shared library (product), the CPU oracle (test infrastructure) and the host executable.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -munsafe-fp-atomics -Wall -Wno-unused-function
CSRC := xpic_amd/csrc
SRCS := $(CSRC)/api.hip $(CSRC)/fields.hip $(CSRC)/particles.hip $(CSRC)/ecsim.hip $(CSRC)/esirkepov.hip $(CSRC)/krylov.hip $(CSRC)/comm.hip
OBJS := $(SRCS:.hip=.o)
HDRS := $(wildcard $(CSRC)/*.h) include/xpic_hip.h

all: xpic_amd/libxpic_hip.so oracle

xpic_amd/libxpic_hip.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(OBJS) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) xpic_amd/libxpic_hip.so
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
