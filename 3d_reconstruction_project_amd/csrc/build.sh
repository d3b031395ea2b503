#!/bin/bash
# Builds lib/libr3d_hip.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [extra hipcc flags]
# Each translation unit is compiled on its own, in parallel.  (An experiment with a second copy of sgm.hip compiled under
# -mllvm -amdgpu-sched-strategy=max-ilp for k_vscan2 alone removed most s_nop wait states from the listing but measured
# equal to the default scheduler when the two builds were interleaved on one box, so it is gone)
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../lib"
obj="$here/../lib/obj"
mkdir -p "$out" "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
COMMON=(--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -Wall -Wno-unused-function "$@")
pids=()
for f in api sgm cloud prepost; do
    "$HIPCC" "${COMMON[@]}" -c "$here/$f.hip" -o "$obj/$f.o" &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$obj"/api.o "$obj"/sgm.o "$obj"/cloud.o "$obj"/prepost.o -o "$out/libr3d_hip.so"
echo "built $out/libr3d_hip.so"
