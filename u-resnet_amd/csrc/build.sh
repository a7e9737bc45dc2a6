#!/bin/bash
# Builds liburesnet_hip.so for gfx950 (MI355X) in-tree: u-resnet_amd/lib/liburesnet_hip.so
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
OBJ="$HERE/../lib/obj"
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${URESNET_EXTRA_FLAGS:-}"
SRCS=(common conv_igemm convt_gemm conv1x1_pw conv1x1_pw_wgrad conv3x3_tile conv3x3_wide wgrad3x3_tile conv_wgrad conv_group bn spatial loss optim)
pids=()
for s in "${SRCS[@]}"; do
    src="$HERE/$s.hip"
    obj="$OBJ/$s.o"
    stale=0
    for dep in "$src" "$HERE"/*.h "$HERE/../../include/uresnet_hip.h"; do
        if [ ! -f "$obj" ] || [ "$dep" -nt "$obj" ]; then stale=1; fi
    done
    if [ "$stale" = 1 ]; then
        ( $HIPCC $FLAGS -c "$src" -o "$obj" ) &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
objs=()
for s in "${SRCS[@]}"; do objs+=("$OBJ/$s.o"); done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/liburesnet_hip.so" "${objs[@]}"
echo "built $OUT/liburesnet_hip.so"
