#!/bin/bash
# Builds lib/libr3d_hip.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [extra hipcc flags]
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    "$@" "$here"/api.hip "$here"/sgm.hip "$here"/cloud.hip "$here"/prepost.hip -o "$out/libr3d_hip.so"
echo "built $out/libr3d_hip.so"
