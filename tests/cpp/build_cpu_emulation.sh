#!/bin/sh
# Builds the host-side emulation of the COLUMN ENGINE: libhgx's own sources for the column path — launch sequences, buffers and
# kernels of hgx_columns.hip, the upload code of hgx_device_image.hip, the host side above them — compiled by g++ against
# tests/cpp/hipshim (device memory = host memory, a launch = a loop over the grid's threads), the liftover engine stubbed out.
# usage: build_cpu_emulation.sh <out.so> [extra compiler flags, e.g. -fsanitize=address,undefined]
# Test infrastructure (tests/test_cpu_emulation.py); libhgx.so is never built this way.
set -e
OUT="$1"; shift
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../hal_amd/csrc"
FLAGS="-O1 -g -std=c++17 -fPIC -pthread -w -DHGX_CPU_EMULATION -I$HERE/hipshim -I$SRC $*"
OBJ="$(dirname "$OUT")/obj_$(basename "$OUT" .so)"
mkdir -p "$OBJ"
pids=""
for f in hgx_columns.hip hgx_device_image.hip; do
    g++ $FLAGS -x c++ -c "$SRC/$f" -o "$OBJ/$f.o" & pids="$pids $!"
done
for f in hgx_capi.cpp hgx_textmem.cpp hgx_comm.cpp hgx_liftover_host.cpp hgx_liftover_text.cpp hgx_blockviz.cpp hgx_columns_host.cpp hgx_image.cpp \
         hgx_mmap_reader.cpp hgx_hdf5_reader.cpp hgx_randgen.cpp; do
    g++ $FLAGS -c "$SRC/$f" -o "$OBJ/$f.o" & pids="$pids $!"
done
g++ $FLAGS -c "$HERE/cpu_liftover_stubs.cpp" -o "$OBJ/stubs.o" & pids="$pids $!"
for p in $pids; do wait $p; done
g++ -shared -fPIC -pthread $* -Wl,-soname,libhgx.so -o "$OUT" "$OBJ"/*.o -ldl
