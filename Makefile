# Top-level build: HIP kernels (gfx950) + host C -> miniasm_amd/lib/libminiasm_amd.so, the miniasm CLI,
# the synthetic PAF generator, the CPU oracle (tests only) and the link-level drop-in check.
#   make            everything
#   make lib        just the library
# hipcc cross-compiles gfx950 without a GPU present.

HIPCC    ?= /opt/rocm/bin/hipcc
CC       ?= gcc
ARCH     ?= gfx950
PKG       = miniasm_amd
CSRC      = $(PKG)/csrc
HOST      = $(PKG)/host
B         = build/obj

# EXTRA: experiment switches, e.g. EXTRA='-DRS_ITEMS=8' (tools/variants.sh builds such libraries next to the product one)
HIPFLAGS  = --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Iinclude -I$(CSRC) $(EXTRA)
CFLAGS    = -O2 -g -Wall -fPIC -Iinclude -I$(HOST) -I$(CSRC)

HIP_SRC   = scan radix hits graph clean ug useq comm paf mahip_api xfer diag
HOST_SRC  = timers name_dict paf_reader ingest_mt ingest_gpu ingest_sharded hits_host graph_host refsort unitig_gfa pipeline sharded
HIP_OBJ   = $(addprefix $(B)/,$(addsuffix .hip.o,$(HIP_SRC)))
HOST_OBJ  = $(addprefix $(B)/,$(addsuffix .o,$(HOST_SRC)))

LIB       = $(PKG)/lib/libminiasm_amd.so
BIN       = $(PKG)/bin/miniasm $(PKG)/bin/pafgen
CORE_TEST = $(PKG)/lib/libma_core_host.so $(PKG)/lib/libclean_host.so

.PHONY: all lib oracle dropin clean
all: lib $(BIN) $(CORE_TEST) oracle dropin

lib: $(LIB)

$(B) $(PKG)/lib $(PKG)/bin:
	mkdir -p $@

$(B)/%.hip.o: $(CSRC)/%.hip $(CSRC)/mahip_internal.hpp $(CSRC)/ma_core.h $(CSRC)/clean_core.h $(CSRC)/ug_core.h include/mahip.h include/miniasm_amd.h | $(B)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(B)/%.o: $(HOST)/%.c $(HOST)/ma_host.h $(HOST)/refsort_body.h include/mahip.h include/miniasm_amd.h | $(B)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIB): $(HIP_OBJ) $(HOST_OBJ) | $(PKG)/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,-Bsymbolic -o $@ $(HIP_OBJ) $(HOST_OBJ) -lz -lm -lpthread -ldl -lrt

$(PKG)/bin/miniasm: $(HOST)/cli.c $(LIB) | $(PKG)/bin
	$(CC) $(CFLAGS) -o $@ $(HOST)/cli.c -L$(PKG)/lib -lminiasm_amd -Wl,-rpath,'$$ORIGIN/../lib' -lz -lm

$(PKG)/bin/pafgen: tools/pafgen.c | $(PKG)/bin
	$(CC) -O2 -Wall -o $@ tools/pafgen.c -lm

# ma_core.h compiled for the host: lets the CPU tests check the per-hit arithmetic against the reference
$(PKG)/lib/libma_core_host.so: tests/core_host.c $(CSRC)/ma_core.h | $(PKG)/lib
	$(CC) -O2 -g -Wall -fPIC -ffp-contract=off -shared -I$(CSRC) -o $@ tests/core_host.c

# clean_core.h / ug_core.h compiled for the host: the cleaners' fixpoint and the unitig construction run on the CPU in the tests
$(PKG)/lib/libclean_host.so: tests/clean_host.cpp $(CSRC)/clean_core.h $(CSRC)/ug_core.h | $(PKG)/lib
	g++ -O2 -g -Wall -fPIC -shared -std=c++17 -I$(CSRC) -o $@ tests/clean_host.cpp

oracle:
	$(MAKE) -C oracle all

# the reference's own driver object linked against OUR library (reference main.c unchanged)
dropin: $(LIB) oracle
	@if [ -f oracle/_ref/main_ref.o ]; then \
	  $(CC) -o oracle/_ref/miniasm_dropin oracle/_ref/main_ref.o -L$(PKG)/lib -lminiasm_amd -Wl,-rpath,'$$ORIGIN/../../$(PKG)/lib' -lz -lm && echo "[dropin] built oracle/_ref/miniasm_dropin"; \
	else echo "[dropin] oracle/_ref/main_ref.o missing; skipped"; fi

clean:
	rm -rf build $(PKG)/lib $(PKG)/bin
	$(MAKE) -C oracle clean
