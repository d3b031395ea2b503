#!/bin/bash
# Issue / wait / LDS / traffic counters of the SGM kernels of one R3D_SGM_IMPL setting (separate --pmc passes, nothing else
# traced; raw output stays in /tmp, the per-kernel means land in gpurun_out/<tag>_pmc_sgm.json).
# Usage (GPU box, repository root): tools/gpu_pmc_sgm.sh <tag> [impl]
set -o pipefail
tag="${1:-r}"; impl="${2:-v2}"
export TMPDIR=/tmp R3D_SGM_IMPL="$impl"
root="$PWD"; out="$root/gpurun_out"; mkdir -p "$out"
cd /tmp
run() {  # run <dir> <counters...>
  d="$1"; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "/tmp/pmc_${tag}_$d" -o p -- python3 "$root/bench.py" --no-torch --no-gicp --no-cpu-baseline --no-extras --steps 4 --warmup 1 --repeats 0 > /dev/null 2> "$out/${tag}_pmc_$d.err" || { echo "pass $d failed"; tail -3 "$out/${tag}_pmc_$d.err"; return 1; }
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 2
run f FETCH_SIZE || exit 3
run w WRITE_SIZE || exit 4
python3 "$root/tools/pmc_summary.py" /tmp/pmc_${tag}_a /tmp/pmc_${tag}_b /tmp/pmc_${tag}_f /tmp/pmc_${tag}_w > "$out/${tag}_pmc_sgm.json"
echo done
