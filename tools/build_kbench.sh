#!/bin/bash
# Builds the standalone kernel A/B harness tools/bin/kbench for gfx950 (development tool; see tools/kbench.hip).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
mkdir -p "$HERE/bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -munsafe-fp-atomics -DEBFI_KBENCH -DEBFI_ABLATE ${KBENCH_FLAGS:-} \
    "$HERE/kbench.hip" "$HERE/../ebfi-be_amd/csrc/runtime.hip" -o "$HERE/bin/kbench"
echo "built $HERE/bin/kbench"
