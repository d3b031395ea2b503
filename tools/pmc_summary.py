"""Summarises rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean of each counter per dispatch.
Usage: python tools/pmc_summary.py <dir> [<dir> ...]  -> JSON on stdout."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                m = re.search(r"\bk_[a-z0-9_]+", name)
                short = m.group(0) if m else name[:40]
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
if os.environ.get("PMC_MEDIAN"):   # robust against one-off launches (e.g. the first, unaligned evaluation of a registration loop)
    out = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in acc.items()}
else:
    out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
print(json.dumps(out, indent=1, sort_keys=True))
