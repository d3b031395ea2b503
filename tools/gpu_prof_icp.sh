#!/bin/bash
# Kernel trace + PMC passes of tools/gpu_bench_gicp.py for two search variants (default packed search, R3D_ICP_IMPL=f32);
# raw rocprofv3 output stays in /tmp on the GPU box, only per-kernel summaries land in gpurun_out/.
# Usage (GPU box, repository root): tools/gpu_prof_icp.sh <tag>
set -eo pipefail
tag="${1:-icp}"
export TMPDIR=/tmp
out="$PWD/gpurun_out"
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
B="TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
for impl in ${IMPLS:-default f32}; do
  export R3D_ICP_IMPL=$impl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_${impl}_kt -o k -- python3 tools/gpu_bench_gicp.py gicp > "$out/${tag}_${impl}_kt.log" 2>&1
  cp /tmp/prof_${tag}_${impl}_kt/k_kernel_stats.csv "$out/${tag}_${impl}_kernel_stats.csv"
  python3 tools/trace_gaps.py /tmp/prof_${tag}_${impl}_kt/k_kernel_trace.csv > "$out/${tag}_${impl}_gaps.json"
  rocprofv3 --pmc $A --output-format csv -d /tmp/prof_${tag}_${impl}_a -o p -- python3 tools/gpu_bench_gicp.py gicp > /dev/null 2>&1
  rocprofv3 --pmc $B --output-format csv -d /tmp/prof_${tag}_${impl}_b -o p -- python3 tools/gpu_bench_gicp.py gicp > /dev/null 2>&1
  PMC_MEDIAN=1 python3 tools/pmc_summary.py /tmp/prof_${tag}_${impl}_a /tmp/prof_${tag}_${impl}_b > "$out/${tag}_${impl}_pmc.json"
done
