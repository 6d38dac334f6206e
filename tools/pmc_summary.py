#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch."""
import csv, glob, re, sys, collections
root = sys.argv[1]


def kname(full):
    """short kernel name; the two launches of k_fast_cells per extraction (levels >= 1 / level 0) stay apart"""
    m = re.search(r"\bk_\w+", full)
    if not m:
        return None
    k = m.group(0)
    t = re.search(r"k_fast_cells<([^>]*)>", full)
    if k == "k_fast_cells" and t:
        k += "<" + t.group(1).replace(" ", "") + ">"
    return k

files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in files:
    for row in csv.DictReader(open(f)):
        k = kname(row["Kernel_Name"]) or row["Kernel_Name"][:40]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("   %-28s n=%-5d mean=%.4g" % (c, n[k][c], acc[k][c] / n[k][c]))
