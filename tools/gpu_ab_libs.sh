#!/bin/bash
# Same-box A/B of several builds of the library: tools/gpu_ab_libs.sh <rounds> <lib-suffix>...   (lib/libr3d_<suffix>.so, alternated;
# prints maps/s and the per-kernel brackets of bench.py for each).  The installed lib/libr3d_hip.so is restored at the end.
set -o pipefail
rounds="$1"; shift
L=3d_reconstruction_project_amd/lib
cp $L/libr3d_hip.so /tmp/r3d_installed.so
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    cp $L/libr3d_$v.so $L/libr3d_hip.so
    python bench.py --no-cpu-baseline --no-c5 --no-extras --no-gicp --repeats 0 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['roofline']['kernel_ms'];print('$v',d['value'],'cost',k['cost'],'hscan',k['hscan'],'vscan',k['vscan_wta'])"
  done
done
cp /tmp/r3d_installed.so $L/libr3d_hip.so
