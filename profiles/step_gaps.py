#!/usr/bin/env python3
"""Timeline of the refiner's per-step kernels from a rocprofv3 kernel-trace CSV: durations, the idle time of the encoder stream between
the last encoder attention of a step and the row kernel of the next, the wait of the tail behind its attention, and the step period.

usage: step_gaps.py kernel_trace.csv"""
import csv
import statistics as st
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    k = "rows" if "enc_rows_kernel" in n else "attn" if "attn_g_kernel" in n else "tail" if "tail_mfma_kernel" in n else None
    if k:
        ev.append((k, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
R = [e for e in ev if e[0] == "rows"]
A = [e for e in ev if e[0] == "attn"]
T = [e for e in ev if e[0] == "tail"]
n = min(len(R), len(A), len(T))
us = lambda x: x / 1e3
per, gap_ra, gap_ar, gap_at, gap_tt, lag = [], [], [], [], [], []
for i in range(n - 1):
    d = R[i + 1][1] - R[i][1]
    if d > 400e3:            # chunk boundary (cache builds in between)
        continue
    per.append(us(d))
    gap_ra.append(us(A[i][1] - R[i][2]))            # row kernel end -> attention start
    gap_ar.append(us(R[i + 1][1] - A[i][2]))        # attention end -> next row kernel start
    gap_at.append(us(T[i][1] - A[i][2]))            # attention end -> tail start
    gap_tt.append(us(T[i + 1][1] - T[i][2]))        # tail end -> next tail start
    if i >= 1:
        lag.append(us(R[i + 1][1] - T[i - 1][2]))   # tail of step k - 1 end -> row kernel of step k + 1 start (it waits for that tail)
dur = lambda E: st.mean(us(e[2] - e[1]) for e in E)
print(f"{n} steps; durations: row kernel {dur(R):.1f} us, attention {dur(A):.1f}, tail {dur(T):.1f}")
print(f"step period (row kernel start to start) mean {st.mean(per):.1f} us, median {st.median(per):.1f}")
for name, g in (("row kernel end -> attention start", gap_ra), ("attention end -> next row kernel start", gap_ar), ("attention end -> tail start", gap_at),
                ("tail end -> next tail start", gap_tt), ("tail(k-1) end -> row kernel(k+1) start", lag)):
    print(f"  {name:42s} mean {st.mean(g):7.1f} us  median {st.median(g):7.1f}  min {min(g):7.1f}  max {max(g):7.1f}")
