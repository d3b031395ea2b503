#!/bin/bash
# One profiling pass of a round on the GPU box (repository root): kernel-trace statistics of the default bench command, then the
# HBM-traffic counters (FETCH_SIZE / WRITE_SIZE, their own passes as MI355X_MICROARCH.md prescribes) of the SGM step and of the
# GICP loop.  Raw rocprofv3 output stays in /tmp; only per-kernel summaries land in gpurun_out/<tag>_*.
# Usage: tools/gpu_profile_round.sh <tag>
set -o pipefail
tag="${1:-r}"
export TMPDIR=/tmp
root="$PWD"
out="$root/gpurun_out"
mkdir -p "$out"
cd /tmp
# 1. kernel statistics.  (a) the bench command with the legs that overlap maps switched off (--no-extras --no-c5: since round 3 the
#    default command also runs three maps in flight and the 8-view batch, whose SGM launches share the chip and would enter the
#    per-kernel averages): this is the summary whose k_hscan2 average must agree with roofline.kernel_ms of the line it printed;
#    (b) the default command as the driver runs it, for the record (all legs; per-kernel averages there mix contended launches)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_kt -o k -- python3 "$root/bench.py" --no-extras --no-c5 > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_kt.err" || exit 1
cp "$(find /tmp/prof_${tag}_kt -name 'k_kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_ktd -o k -- python3 "$root/bench.py" > "$out/${tag}_bench_default_under_rocprof.json" 2> "$out/${tag}_ktd.err" || echo "default command under rocprofv3 failed: $(tail -1 "$out/${tag}_ktd.err")" > "$out/${tag}_kt.note"
cp "$(find /tmp/prof_${tag}_ktd -name 'k_kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats_default_command.csv" 2>/dev/null
# 2. HBM traffic of the SGM kernels (counters in their own passes, nothing else traced)
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_${tag}_f -o p -- python3 "$root/bench.py" --no-torch --no-gicp --no-cpu-baseline --steps 6 --warmup 2 --repeats 0 > /dev/null 2> "$out/${tag}_pmc_f.err" || exit 2
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_${tag}_w -o p -- python3 "$root/bench.py" --no-torch --no-gicp --no-cpu-baseline --steps 6 --warmup 2 --repeats 0 > /dev/null 2> "$out/${tag}_pmc_w.err" || exit 3
python3 "$root/tools/pmc_summary.py" /tmp/prof_${tag}_f /tmp/prof_${tag}_w > "$out/${tag}_traffic_sgm.json"
# 3. the same for the registration loop (1 M + 1 M points, GICP); median per kernel: the first evaluation of the unaligned pair is atypical
R3D_NO_TORCH_PRELOAD=1 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_${tag}_gf -o p -- python3 "$root/tools/gpu_bench_gicp.py" gicp > "$out/${tag}_pmc_gicp.log" 2>&1 || exit 4
R3D_NO_TORCH_PRELOAD=1 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_${tag}_gw -o p -- python3 "$root/tools/gpu_bench_gicp.py" gicp >> "$out/${tag}_pmc_gicp.log" 2>&1 || exit 5
PMC_MEDIAN=1 python3 "$root/tools/pmc_summary.py" /tmp/prof_${tag}_gf /tmp/prof_${tag}_gw > "$out/${tag}_traffic_gicp.json"
echo done
