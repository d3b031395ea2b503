#!/bin/bash
# A/B of the two cell-table kinds (R3D_CELL_TABLE = dense | lines) for the kNN kernels and the registration loop, interleaved on
# one box.  Usage (GPU box, repository root): tools/gpu_ab_table.sh > gpurun_out/<tag>.log
for rep in 1 2; do
  for t in dense lines; do
    echo "== $t (rep $rep)"
    R3D_CELL_TABLE=$t python3 tools/gpu_bench_normals.py 20 2>/dev/null | tail -1
    R3D_CELL_TABLE=$t python3 tools/gpu_bench_normals.py 30 0.02 2>/dev/null | tail -1
    R3D_CELL_TABLE=$t python3 tools/gpu_bench_gicp.py gicp 2>/dev/null | tail -2
  done
done
