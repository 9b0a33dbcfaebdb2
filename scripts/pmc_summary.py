"""Summarise the CSVs written by scripts/pmc_passes.sh: per kernel name, sum of every counter over dispatches."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (row["Dispatch_Id"], row["Counter_Name"])
        if key not in seen:
            seen.add(key)
            calls[k][row["Counter_Name"]] += 1
for k in tot:
    if "mipt" not in k:
        continue
    print(k)
    for c, v in sorted(tot[k].items()):
        print(f"  {c:34s} {v:18.0f}   dispatches {calls[k][c]}")
