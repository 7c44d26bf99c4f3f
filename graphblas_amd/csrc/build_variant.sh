#!/bin/bash
# Tuning build of one variant of the saxpy3 kernels:  build_variant.sh NAME [extra compiler flags]
# -> ../ab/libgba_NAME.so (PLUS_TIMES_FP64 only, -DGBA_DEV_FAST); the other translation units are compiled once
# and shared between variants.  Run with GBA_LIB_PATH=<that file>.
set -e
cd "$(dirname "$0")"
NAME=$1; shift
B=${TMPDIR:-/tmp}/gba_build_variant
mkdir -p "$B" ../ab
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -DGBA_DEV_FAST -w"
for f in gba_runtime gba_dense gba_dot3 gba_meta gba_util gba_jit; do
    if [ ! -f "$B/$f.o" ] || [ "$f.hip" -nt "$B/$f.o" ] || [ gba_internal.hpp -nt "$B/$f.o" ] || [ gba_ops.hpp -nt "$B/$f.o" ]; then
        /opt/rocm/bin/hipcc $F -c $f.hip -o "$B/$f.o" &
    fi
done
/opt/rocm/bin/hipcc $F "$@" -c gba_saxpy3.hip -o "$B/gba_saxpy3_$NAME.o"
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 "$B"/gba_{runtime,dense,dot3,meta,util,jit}.o "$B/gba_saxpy3_$NAME.o" -L/opt/rocm/lib -lhiprtc -o ../ab/libgba_$NAME.so
