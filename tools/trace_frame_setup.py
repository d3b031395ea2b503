"""Kernel timeline of the set-up of the LAST frame of a scanning loop from a rocprofv3 --kernel-trace CSV: every launch between
the previous frame's last k_icp_step and this frame's first k_icp_eval (start offset, duration, gap before; us), and the sums
by kernel.  Usage: python tools/trace_frame_setup.py <kernel_trace.csv>"""
import collections, csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\bk_[a-z0-9_]+", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:32]))
rows.sort()
last_eval = max(i for i, x in enumerate(rows) if x[2].startswith("k_icp_eval"))
i = last_eval
while i > 0 and rows[i - 1][2].startswith("k_icp"):       # first evaluation of the last registration
    i -= 1
first_eval = i
j = first_eval - 1
while j > 0 and not rows[j - 1][2].startswith("k_icp"):   # back to the end of the previous frame's loop
    j -= 1
t0 = rows[j][0]
by = collections.Counter()
for k in range(j, first_eval + 1):
    s, e, n = rows[k]
    gap = (s - rows[k - 1][1]) / 1e3 if k > j else 0.0
    by[n] += (e - s) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  {n:24s} {(e - s) / 1e3:8.1f} us  gap {gap:7.1f}")
print(f"set-up span {(rows[first_eval][0] - t0) / 1e3:.1f} us over {first_eval - j} launches; busy {sum(v for k_, v in by.items() if not k_.startswith('k_icp')):.1f} us")
for n, v in by.most_common(12):
    print(f"   {n:24s} {v:8.1f} us")
