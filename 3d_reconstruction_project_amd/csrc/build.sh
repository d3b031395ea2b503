#!/bin/bash
# Builds lib/libr3d_hip.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [extra hipcc flags]
# Each translation unit is compiled on its own (in parallel) so that per-file code-generation options are possible:
# sgm.hip is compiled a second time with -DR3D_TU_VSCAN (only k_vscan2 + its launcher) under the ILP-oriented machine
# scheduler, which fills the wait state gfx950 needs after a packed (VOP3P) result with independent work instead of
# s_nop (k_vscan2: 240 -> 47 s_nop per unrolled loop body, 0.84 -> 0.79 ms); the other SGM kernels are slower under it
# (k_cost2 0.68 -> 0.87 ms: 145 instead of 94 VGPRs).  R3D_VSCAN_SCHED overrides the strategy for A/B runs.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../lib"
obj="$here/../lib/obj"
mkdir -p "$out" "$obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
COMMON=(--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -Wall -Wno-unused-function "$@")
VSCAN_SCHED="${R3D_VSCAN_SCHED:-max-ilp}"
pids=()
for f in api sgm cloud prepost; do
    "$HIPCC" "${COMMON[@]}" -c "$here/$f.hip" -o "$obj/$f.o" &
    pids+=($!)
done
"$HIPCC" "${COMMON[@]}" -DR3D_TU_VSCAN -mllvm "-amdgpu-sched-strategy=$VSCAN_SCHED" -c "$here/sgm.hip" -o "$obj/sgm_vscan.o" &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$obj"/api.o "$obj"/sgm.o "$obj"/sgm_vscan.o "$obj"/cloud.o "$obj"/prepost.o -o "$out/libr3d_hip.so"
echo "built $out/libr3d_hip.so"
