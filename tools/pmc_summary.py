#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch."""
import csv, glob, re, sys, collections
root = sys.argv[1]
files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in files:
    for row in csv.DictReader(open(f)):
        m = re.search(r"\bk_\w+", row["Kernel_Name"])
        k = m.group(0) if m else row["Kernel_Name"][:40]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("   %-28s n=%-5d mean=%.4g" % (c, n[k][c], acc[k][c] / n[k][c]))
