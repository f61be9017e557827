#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV: per-kernel totals for the LAST V-cycle-sized window and the idle gaps."""
import csv
import collections
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 5
# the last `1/ncyc` of the dispatches after set-up: find V-cycle boundaries = the level-1 residual kernel launches
res = [(i, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_residual")]
mx = max(d for _, d in res)
big = [i for i, d in res if d > 0.7 * mx]  # the level-1 residual launches
lo, hi = big[-2] + 0, big[-1]  # one full cycle between two level-1 residuals
sel = rows[lo:hi]
tot = collections.defaultdict(lambda: [0, 0])
busy = 0
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = r["Kernel_Name"].split("(")[0][:48]
    tot[k][0] += d; tot[k][1] += 1
    busy += d
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
print(f"window: {len(sel)} dispatches, span {span/1e3:.1f} us, busy {busy/1e3:.1f} us, idle {(span-busy)/1e3:.1f} us")
for k, (d, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:50s} n={n:4d} total={d/1e3:9.1f} us  avg={d/n/1e3:8.2f} us")
