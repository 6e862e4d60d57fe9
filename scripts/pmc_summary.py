#!/usr/bin/env python3
"""Per-kernel mean of every counter in a rocprofv3 counter_collection CSV."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "?")
        acc[k][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", "0") or 0))
for k, ctrs in acc.items():
    if "fsea" not in k:
        continue
    for c, vals in sorted(ctrs.items()):
        print("%s %s mean=%.6g n=%d" % (k, c, sum(vals) / len(vals), len(vals)))
