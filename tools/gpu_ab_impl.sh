#!/bin/bash
# Interleaved A/B of two R3D_SGM_IMPL settings on ONE GPU box: bench.py (no CPU legs) alternately with each, REPS times;
# prints value (one map in flight), pipelined (three in flight) and the per-kernel milliseconds.
# Usage (GPU box, repository root): tools/gpu_ab_impl.sh v2 v4 [REPS]
set -euo pipefail
A="$1"; B="$2"; REPS="${3:-3}"
for rep in $(seq "$REPS"); do
  for v in "$A" "$B"; do
    R3D_SGM_IMPL=$v python bench.py --no-cpu-baseline --no-gicp --no-c5 --repeats 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'value', d['value'], 'pipelined', d.get('pipelined', {}).get('value'), d['roofline']['kernel_ms'])"
  done
done
