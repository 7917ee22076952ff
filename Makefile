# Convenience targets; the driver itself uses __graft_entry__.build(), pytest and bench.py directly.
PY ?= python

.PHONY: build test-cpu test-gpu bench c-host clean

build:            ## hipcc --offload-arch=gfx950 -> galois_amd/libgalois_amd.so (+ the oracle's C library)
	$(PY) -c "import __graft_entry__ as g; g.build()"

test-cpu:         ## oracle vs golden vectors / live reference, host logic, C-ABI exports, gloo exchange
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu:         ## parity of every kernel through the C-ABI (needs an MI355X)
	$(PY) -m pytest tests -x -q -m gpu

bench:            ## one JSON line: GF(2^8) multiply, 1e8 elements, roofline + CPU baseline
	$(PY) bench.py

c-host:           ## the plain-C host example over include/galois_amd.h
	gcc -std=c99 -Wall -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_host_rs.c \
	    -Lgalois_amd -lgalois_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$(CURDIR)/galois_amd -Wl,-rpath,/opt/rocm/lib -o c_host_rs

clean:
	rm -rf galois_amd/_obj galois_amd/libgalois_amd.so oracle/_build c_host_rs
