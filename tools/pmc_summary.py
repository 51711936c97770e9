#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: calls, mean counter value per dispatch.

    python tools/pmc_summary.py <dir written by rocprofv3 -d> <out.json> [name filter ...]
"""
import csv
import glob
import json
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
filters = sys.argv[3:] or ["af::", "supp_", "chain_", "resample"]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for path in glob.glob(f"{src}/**/*counter_collection.csv", recursive=True):
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if not any(f in name for f in filters):
                continue
            cell = acc[name][row["Counter_Name"]]
            cell[0] += 1
            cell[1] += float(row["Counter_Value"])
summary = {name: {counter: {"dispatches": n, "mean_per_dispatch": total / n} for counter, (n, total) in counters.items()}
           for name, counters in acc.items()}
json.dump(summary, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True)[:3000])
