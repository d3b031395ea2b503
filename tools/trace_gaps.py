"""Timeline of the registration loop from a rocprofv3 --kernel-trace CSV: per launch of k_icp_eval / k_icp_step the
duration and the idle gap before it.  Usage: python tools/trace_gaps.py <kernel_trace.csv>  -> JSON on stdout."""
import csv
import json
import statistics
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = {}
prev_end = None
for r in rows:
    name = r["Kernel_Name"]
    short = "eval" if "k_icp_eval" in name else "step" if "k_icp_step" in name else None
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if short and prev_end is not None:
        d = out.setdefault(short, {"dur_us": [], "gap_before_us": []})
        d["dur_us"].append((e - s) / 1e3)
        d["gap_before_us"].append((s - prev_end) / 1e3)
    prev_end = e
res = {}
for k, d in out.items():
    res[k] = {"launches": len(d["dur_us"]), "dur_us_median": round(statistics.median(d["dur_us"]), 2),
              "dur_us_max": round(max(d["dur_us"]), 1), "gap_before_us_median": round(statistics.median(d["gap_before_us"]), 2),
              "gap_before_us_p90": round(sorted(d["gap_before_us"])[int(0.9 * len(d["gap_before_us"]))], 2)}
print(json.dumps(res, indent=1))
