#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV by (kernel, grid): calls, total ms, average us."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    key = (r["Kernel_Name"][:48], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[key][0] += 1
    agg[key][1] += d
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot / 1e6:.1f} ms over {len(rows)} launches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{k[0]:50s} grid=({k[1]},{k[2]}) calls={v[0]:6d} total={v[1] / 1e6:9.2f} ms avg={v[1] / v[0] / 1e3:9.1f} us")
