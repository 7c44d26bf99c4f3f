#!/bin/bash
# Tuning build: the library with the in-kernel stopwatch of the panel kernels (-DGBA_PANEL_TICKS),
# PLUS_TIMES_FP64 only (-DGBA_DEV_FAST; GBA_TICKFLAG= leaves the stopwatch out, GBA_EXTRA adds flags), as ../libgba_ticks.so; run with GBA_LIB_PATH=<that file> GBA_PANEL_PROF=1.
set -e
cd "$(dirname "$0")"
B=${TMPDIR:-/tmp}/gba_build_ticks
mkdir -p "$B"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics ${GBA_TICKFLAG--DGBA_PANEL_TICKS} -DGBA_DEV_FAST $GBA_EXTRA -w"
for f in gba_runtime gba_saxpy3 gba_dense gba_dot3 gba_meta gba_util gba_jit; do
    /opt/rocm/bin/hipcc $F -c $f.hip -o "$B/$f.o" &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 "$B"/*.o -L/opt/rocm/lib -lhiprtc -o ../libgba_ticks.so
