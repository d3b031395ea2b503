#!/bin/bash
# Kernel trace of the set-up paths (grid build / cell sort / voxel grid) behind one C5 view chain and one 1 M-point registration:
# per-kernel summary + the ordered launch list of ONE call with start offsets, so gaps (host round trips) show.
# Usage (GPU box, repository root): tools/gpu_trace_setup.sh <tag>
set -eo pipefail
tag="${1:-setup}"
export TMPDIR=/tmp
out="$PWD/gpurun_out"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag} -o k -- python3 tools/gpu_bench_view_resident.py > "$out/${tag}_view.log" 2>&1
cp /tmp/prof_${tag}/k_kernel_stats.csv "$out/${tag}_view_kernel_stats.csv"
python3 - "$out/${tag}_view_timeline.txt" /tmp/prof_${tag}/k_kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last occurrence of the disparity flags kernel starts the last cloud chain
last = max(i for i, r in enumerate(rows) if "k_disp_flags" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
with open(sys.argv[1], "w") as f:
    prev_end = t0
    for r in rows[last:last + 80]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        import re
        name = re.sub(r"\(anonymous namespace\)::|void |rocprim::ROCPRIM_\d+_NS::detail::", "", r["Kernel_Name"]).split("(")[0][:70]
        f.write(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name}\n")
        prev_end = e
PY
# second part: set-up of a 1 M-point registration (grid build of the target, Morton sort of the source) -- the kernels in front of
# the last k_pack_q10 of tools/gpu_bench_gicp.py
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_icp -o k -- python3 tools/gpu_bench_gicp.py gicp > "$out/${tag}_icp.log" 2>&1
cp /tmp/prof_${tag}_icp/k_kernel_stats.csv "$out/${tag}_icp_kernel_stats.csv"
python3 - "$out/${tag}_icp_timeline.txt" /tmp/prof_${tag}_icp/k_kernel_trace.csv <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[2])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_pack_q10" in r["Kernel_Name"])
first = max(0, last - 40)
t0 = int(rows[first]["Start_Timestamp"])
with open(sys.argv[1], "w") as f:
    prev_end = t0
    for r in rows[first:last + 30]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = re.sub(r"\(anonymous namespace\)::|void |rocprim::ROCPRIM_\d+_NS::detail::", "", r["Kernel_Name"]).split("(")[0][:70]
        f.write(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name}\n")
        prev_end = e
PY
