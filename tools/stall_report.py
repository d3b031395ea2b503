"""Lists the HIP API calls longer than a threshold from a rocprofv3 --hip-trace CSV, with the GPU activity (kernel trace) inside
each: tells a host-side wait on a busy GPU from a wait on an IDLE one (a wake-up that came late).  Usage:
python tools/stall_report.py <dir with *_hip_api_trace.csv and *_kernel_trace.csv> [threshold_ms]"""
import csv, glob, json, os, sys
d = sys.argv[1]
thr = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
api = glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)
ker = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
kern = []
for f in ker:
    for r in csv.DictReader(open(f)):
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
kern.sort()
rows, hist = [], {}
for f in api:
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        hist.setdefault(r["Function"], []).append(e - s)
        if e - s >= thr:
            busy = sum(max(0, min(e, ke) - max(s, ks)) for ks, ke in kern if ke > s and ks < e)
            last_end = max([ke for ks, ke in kern if ke <= e and ke >= s] or [s])
            rows.append({"function": r["Function"], "ms": round((e - s) / 1e6, 3), "gpu_busy_ms_inside": round(busy / 1e6, 3),
                         "ms_between_last_kernel_end_and_return": round((e - last_end) / 1e6, 3)})
summary = {k: {"calls": len(v), "median_us": round(sorted(v)[len(v) // 2] / 1e3, 1), "max_ms": round(max(v) / 1e6, 3)} for k, v in hist.items()}
print(json.dumps({"threshold_ms": thr / 1e6, "long_calls": rows, "per_function": summary}, indent=1))
