#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, per counter: mean value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "?")
            short = name.split("(")[0].replace("void ", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    if "rocclr" in k:
        continue
    print(f"== {k}")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
