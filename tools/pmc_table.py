#!/usr/bin/env python3
"""average every collected PMC counter per kernel from rocprofv3 --output-format csv directories"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("calm::", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel | " + " | ".join(names))
for k in sorted(acc):
    if not k.startswith("k_"):
        continue
    print(k + " | " + " | ".join(f"{sum(acc[k][c])/len(acc[k][c]):.0f}" if c in acc[k] else "-" for c in names))
