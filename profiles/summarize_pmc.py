#!/usr/bin/env python3
"""Average PMC counter values per kernel from a rocprofv3 --pmc counter_collection CSV."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:44]
    c = agg[k][r["Counter_Name"]]
    c[0] += 1
    c[1] += float(r["Counter_Value"])
want = sys.argv[2:] or None
for k, cs in agg.items():
    if want and not any(w in k for w in want):
        continue
    print(k)
    for name, (n, tot) in sorted(cs.items()):
        print(f"    {name:34s} n={n:5d} avg={tot / n:16.1f}")
