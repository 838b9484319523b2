#!/usr/bin/env python
"""Aggregate rocprofv3 counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "?")[:70]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in acc.items():
    n = max(len(v) for v in cs.values())
    print(f"== {name}  ({n} dispatches)")
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} mean {sum(v)/len(v):16.1f}")
