#!/usr/bin/env python3
"""Median duration per (kernel, grid size) from a rocprofv3 kernel trace: python3 scripts/vcycle_trace.py <kernel_trace.csv> [n]"""
import collections
import csv
import sys

d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d[(n, r.get("Grid_Size_X") or r.get("Grid_Size"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
    v = sorted(v)
    print(f"{k[0][:52]:52s} grid {k[1]:>8s} n={len(v):4d} med {v[len(v)//2]/1e3:8.1f} us min {v[0]/1e3:8.1f} sum {sum(v)/1e6:7.2f} ms")
