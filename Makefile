# Builds the gfx950 shared library (product), the CPU oracle (test infrastructure) and the host executable.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -munsafe-fp-atomics -Wall -Wno-unused-function $(EXTRA)
CSRC := xpic_amd/csrc
SRCS := $(CSRC)/api.hip $(CSRC)/fields.hip $(CSRC)/particles.hip $(CSRC)/ecsim.hip $(CSRC)/ecsim_ws.hip $(CSRC)/esirkepov.hip $(CSRC)/krylov.hip $(CSRC)/precond.hip $(CSRC)/comm.hip $(CSRC)/eccapfim.hip
OBJS := $(SRCS:.hip=.o)
HDRS := $(wildcard $(CSRC)/*.h) include/xpic_hip.h

all: xpic_amd/libxpic_hip.so xpic_amd/host/xpic_hip.out oracle

xpic_amd/libxpic_hip.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(OBJS) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

HOST := xpic_amd/host
xpic_amd/host/xpic_hip.out: $(HOST)/main.cpp $(HOST)/xpic_host.cpp $(HOST)/xpic_host.h $(HOST)/json.h include/xpic_hip.h xpic_amd/libxpic_hip.so
	g++ -O2 -std=c++17 -Wall -Wextra -o $@ $(HOST)/main.cpp $(HOST)/xpic_host.cpp -Lxpic_amd -lxpic_hip -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath-link,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) xpic_amd/libxpic_hip.so xpic_amd/host/xpic_hip.out
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
