"""Launch-by-launch timeline of the LAST registration loop in a rocprofv3 --kernel-trace CSV: start offset, duration and the
idle gap before every k_icp_* launch (us).  Usage: python tools/trace_icp_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\bk_[a-z0-9_]+", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:25]))
rows.sort()
icp = [i for i, x in enumerate(rows) if x[2].startswith("k_icp")]
# the last run of k_icp launches without a gap above 1 ms
end = icp[-1]
start = end
while start > 0 and rows[start - 1][2].startswith("k_icp") and rows[start][0] - rows[start - 1][1] < 1_000_000:
    start -= 1
t0 = rows[start][0]
busy = 0
for i in range(start, end + 1):
    s, e, n = rows[i]
    gap = (s - rows[i - 1][1]) / 1e3 if i > start else 0.0
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  {n:14s} {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}")
print(f"span {(rows[end][1] - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, launches {end - start + 1}")
