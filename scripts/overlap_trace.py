#!/usr/bin/env python3
"""How much of the halo exchanges runs beside a smoother kernel: rocprofv3 --kernel-trace CSV of ONE rank of a multi-rank run.
For every k_halo_exchange dispatch: its duration and the part of it during which a colour-pass kernel (k_relax_nz / k_relax_ks / k_relax_tall) of
the same process was executing on another queue; per kernel name the queues (streams) it ran on.  python3 scripts/overlap_trace.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
qcol = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
name = lambda r: r["Kernel_Name"].replace("void ", "").split("(")[0]
relax = [r for r in rows if name(r).startswith(("k_relax_nz", "k_relax_ks<", "k_relax_tall"))]
halo = [r for r in rows if name(r).startswith("k_halo_exchange")]
queues = defaultdict(set)
for r in rows:
    if qcol:
        queues[name(r)[:40]].add(r[qcol])
print(f"{len(rows)} dispatches, {len(halo)} k_halo_exchange, {len(relax)} colour passes; queue column: {qcol}")
for k in ("k_halo_exchange", "k_relax_nz", "k_relax_ks<"):
    qs = sorted({q for n, s in queues.items() if n.startswith(k) for q in s})
    print(f"  {k:18s} ran on queues {qs}")
# overlap of each exchange with colour passes on ANOTHER queue
j0 = 0
tot = cov = 0
big = []
for h in halo:
    c = 0
    while j0 < len(relax) and relax[j0]["e"] < h["s"] - 10_000_000:
        j0 += 1
    for r in relax[j0:]:
        if r["s"] > h["e"]:
            break
        if qcol and r[qcol] == h[qcol]:
            continue
        c += max(0, min(r["e"], h["e"]) - max(r["s"], h["s"]))
    d = h["e"] - h["s"]
    tot += d
    cov += min(c, d)
    big.append((d, min(c, d)))
if halo:
    print(f"exchange time total {tot / 1e3:.1f} us over {len(halo)} dispatches (avg {tot / len(halo) / 1e3:.2f} us); "
          f"beside a colour pass of another queue: {cov / 1e3:.1f} us = {100.0 * cov / tot:.1f} %")
    big.sort(reverse=True)
    n10 = max(1, len(big) // 10)
    print(f"longest tenth of the exchanges: avg {sum(d for d, _ in big[:n10]) / n10 / 1e3:.2f} us, {100.0 * sum(c for _, c in big[:n10]) / sum(d for d, _ in big[:n10]):.1f} % beside a colour pass")
