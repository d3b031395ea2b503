#!/bin/bash
# Interleaved A/B of the fused evaluation kernel (k_icp_eval) against the search / accumulate split (R3D_ICP_SPLIT=1) on ONE
# box: the 1 M + 1 M GICP loop of tools/gpu_bench_gicp.py, REPS alternations; then kernel statistics and the counters the
# round-3 verdict named (SQ_WAVES, SQ_INSTS_VALU, SQ_WAIT_INST_ANY, TCC_HIT / TCC_MISS, FETCH_SIZE, WRITE_SIZE) for both.
# Usage (GPU box, repository root): tools/gpu_ab_icp_split.sh <tag> [REPS]
set -o pipefail
tag="${1:-icp_split}"; REPS="${2:-3}"
export TMPDIR=/tmp R3D_NO_TORCH_PRELOAD=1
root="$PWD"; out="$root/gpurun_out"; mkdir -p "$out"
log="$out/${tag}_ab.log"; : > "$log"
for rep in $(seq "$REPS"); do
  for sp in 0 1; do
    echo "== R3D_ICP_SPLIT=$sp (alternation $rep)" >> "$log"
    R3D_ICP_SPLIT=$sp python3 tools/gpu_bench_gicp.py gicp 2>&1 | grep "^gicp" >> "$log"
  done
done
cd /tmp
for sp in 0 1; do
  export R3D_ICP_SPLIT=$sp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_${sp}_kt -o k -- python3 "$root/tools/gpu_bench_gicp.py" gicp > /dev/null 2>&1
  cp "$(find /tmp/prof_${tag}_${sp}_kt -name 'k_kernel_stats.csv' | head -1)" "$out/${tag}_split${sp}_kernel_stats.csv"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/prof_${tag}_${sp}_a -o p -- python3 "$root/tools/gpu_bench_gicp.py" gicp > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/prof_${tag}_${sp}_b -o p -- python3 "$root/tools/gpu_bench_gicp.py" gicp > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_${tag}_${sp}_f -o p -- python3 "$root/tools/gpu_bench_gicp.py" gicp > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_${tag}_${sp}_w -o p -- python3 "$root/tools/gpu_bench_gicp.py" gicp > /dev/null 2>&1
  PMC_MEDIAN=1 python3 "$root/tools/pmc_summary.py" /tmp/prof_${tag}_${sp}_a /tmp/prof_${tag}_${sp}_b /tmp/prof_${tag}_${sp}_f /tmp/prof_${tag}_${sp}_w > "$out/${tag}_split${sp}_pmc.json"
done
cat "$log"
