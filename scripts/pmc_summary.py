#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter_collection CSVs (one pass FETCH_SIZE, one pass WRITE_SIZE) into the per-launch
HBM traffic of the dominant kernels, with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM):
FETCH_SIZE is in KiB and reports 1/2 of a wide coalesced streaming read (calibrated here on k_fine2coarse, which
reads every fine r exactly once); WRITE_SIZE (KiB) is exact.
  python3 scripts/pmc_summary.py <fetch.csv> <write.csv> <cells> > profiles/rNN_pmc_traffic.json"""
import csv
import json
import sys


def top(path, counter):
    best = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        v = float(r["Counter_Value"])
        if v > best.get(k, 0.0):
            best[k] = v
    return best


fetch, write = top(sys.argv[1], "FETCH_SIZE"), top(sys.argv[2], "WRITE_SIZE")
cells = int(sys.argv[3])
out = {"units": "bytes per launch of the level-1 (largest) dispatch of each kernel", "cells": cells,
       "fetch_correction": 2.0, "kernels": {}}
f2c = fetch.get("k_fine2coarse")
if f2c:
    out["calibration"] = {"kernel": "k_fine2coarse", "FETCH_SIZE_KiB": f2c, "bytes_actually_read": cells * 8,
                          "ratio": cells * 8 / (f2c * 1024)}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith(("k_relax", "k_residual", "k_coarse2fine", "k_fine2coarse", "k_rbseq", "k_restrict_chain")):
        continue
    fb, wb = 2.0 * fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    out["kernels"][k] = {"FETCH_SIZE_KiB_raw": fetch.get(k), "WRITE_SIZE_KiB_raw": write.get(k),
                         "read_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb}
print(json.dumps(out, indent=1))
