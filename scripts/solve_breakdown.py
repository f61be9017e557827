#!/usr/bin/env python3
"""Per-kernel time of the solve_p iterations in a rocprofv3 --kernel-trace CSV (the last `nit` iterations = everything after the last
k_sumsq), and the idle gaps between consecutive kernels.  python3 scripts/solve_breakdown.py <kernel_trace.csv> <nit> [bygrid]
(bygrid: one line per kernel AND launch size, which separates the levels of the hierarchy)"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
nit = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_sumsq"))
rows = rows[last:]
tot, cnt = defaultdict(float), defaultdict(int)
gap = 0.0
for a, b in zip(rows, rows[1:]):
    gap += max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if len(sys.argv) > 3:
        n = "%s  grid %sx%sx%s" % (n[:60], r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
    tot[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[n] += 1
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"span per iteration {span / nit / 1e3:.1f} us, busy {sum(tot.values()) / nit / 1e3:.1f} us, gaps {gap / nit / 1e3:.1f} us, launches per iteration {len(rows) / nit:.1f}")
for n, t in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{t / nit / 1e3:9.1f} us/it  {cnt[n] / nit:6.1f} calls/it  {t / cnt[n] / 1e3:8.2f} us avg  {n[:90]}")
