#!/usr/bin/env python3
"""Which kernels do MIOpen's zero fills (SubTensorOpWithScalar1d) precede, and how long are they?  Reads a rocprofv3 kernel_trace.csv
of bench.py and attributes every fill to the next non-fill kernel on its stream; prints time per (following kernel) for the last steps."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(len(rows) * 0.4):]            # steady state
agg = defaultdict(lambda: [0, 0.0, 0.0])
pending = []
for r in rows:
    name = r["Kernel_Name"]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "SubTensorOpWithScalar" in name:
        pending.append(dur)
        continue
    if pending:
        key = name.split("(")[0][:70]
        a = agg[key]
        a[0] += len(pending); a[1] += sum(pending); a[2] = max(a[2], max(pending))
        pending = []
tot = sum(a[1] for a in agg.values())
print(f"fills attributed: {sum(a[0] for a in agg.values())} launches, {tot / 1e3:.3f} ms over the window")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{a[1]:10.1f} us  {a[0]:5d} fills  max {a[2]:6.1f} us  before {k}")
