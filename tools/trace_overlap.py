"""Reads a rocprofv3 --kernel-trace CSV and prints, for the last map in it, every SGM kernel launch with start / end relative to
the map's first kernel (us) and the queue it ran on: shows whether the cost slabs and the forward-scan launches really overlap."""
import csv, re, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        m = re.search(r"\bk_[a-z0-9_]+", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:30], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] == "k_prefilter"]
a = starts[-2] if len(starts) > 1 else starts[-1]
b = starts[-1] if len(starts) > 1 else len(rows)
t0 = rows[a][0]
for s, e, n, q in rows[a:b]:
    print(f"{n:14s} q{q:>3s} start {(s - t0) / 1e3:9.1f}  end {(e - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}")
