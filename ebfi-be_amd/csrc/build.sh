#!/bin/bash
# Builds libebfi_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
# EBFI_LIB_OUT / EBFI_OBJ_DIR: a second build beside the product one (ablation flags via EBFI_EXTRA_FLAGS, A/B runs with
# EBFI_LIB_PATH); the defaults are the in-tree product library
OUTLIB="${EBFI_LIB_OUT:-$OUT/libebfi_hip.so}"
OBJ="${EBFI_OBJ_DIR:-$HERE/obj}"
if [ -n "${EBFI_EXTRA_FLAGS:-}" ] && { [ -z "${EBFI_LIB_OUT:-}" ] || [ -z "${EBFI_OBJ_DIR:-}" ]; }; then
    echo "build.sh: EBFI_EXTRA_FLAGS changes the kernels: build into a separate file (set EBFI_LIB_OUT and EBFI_OBJ_DIR)" >&2
    exit 2
fi
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics ${EBFI_EXTRA_FLAGS:-}"
objs=()
pids=()
for src in "$HERE"/*.hip; do
    obj="$OBJ/$(basename "${src%.hip}").o"
    objs+=("$obj")
    stale=0
    for dep in "$src" "$HERE"/*.hpp "$HERE/../../include/ebfi_hip.h"; do   # (every header of csrc/: conv2d_f16.inc.hpp is one)
        [ "$dep" -nt "$obj" ] && stale=1
    done
    if [ ! -f "$obj" ] || [ "$stale" = 1 ]; then
        $HIPCC $FLAGS -c "$src" -o "$obj" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUTLIB" "${objs[@]}"
echo "built $OUTLIB"
