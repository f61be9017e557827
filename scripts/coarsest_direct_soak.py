#!/usr/bin/env python3
"""Soak test of k_coarse_direct's last-workgroup reduction (mgx_relax_coarse.hip: partial sums released by a fence + ticket, acquired by the last
arriver of a row tile; the workgroups of a launch sit on all eight XCDs): N repetitions of three solve_p iterations (18 direct coarsest solves
each) with the reference default smoother, every word of the finest AND of the coarsest level's p compared with the first repetition's (the sum is
formed in a fixed order: the same bits every time) and, once, with the sweeps' result (1e-12), while a second stream keeps the memory system busy
with device-to-device copies of varying size.
python3 scripts/coarsest_direct_soak.py [reps [nx ny nz]] [--json path]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mgroms_amd as mg  # noqa: E402
from mgroms_amd import nhydro  # noqa: E402
from mgroms_amd.testcases import seamount_geometry, resting_column_state  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
reps = int(args[0]) if args else 200
jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
nx, ny, nz = (int(a) for a in args[1:4]) if len(args) >= 4 else (512, 512, 64)
torch.cuda.set_device(0)
nhydro.set_verbose(0)


def start(direct):
    nhydro.set_option("coarsest_direct", direct)
    mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method="RB"))
    mg.nhydro_matrices(*seamount_geometry(nx, ny, 1, 1, 0), None, 4e3, 0.0, 0.0)
    nhydro.compute_rhs(*resting_column_state(nx, ny, nz))


start(0)
n, hist_sweeps = mg.solve_p(1e-30, 3)
p_sweeps = mg.grid(1).p.copy()
mg.nhydro_clean()

start(1)
nl = mg.nlevs()
side = torch.cuda.Stream()
a = torch.empty(64 << 20, dtype=torch.float64, device="cuda")   # 512 MB
b = torch.empty_like(a)
bad, t0, ref, refc, hist0 = 0, time.time(), None, None, None
rng = np.random.default_rng(7)
d0 = nhydro.get_option("coarsest_direct_solves")
for rep in range(reps):
    with torch.cuda.stream(side):
        for _ in range(int(rng.integers(1, 6))):
            m = int(rng.integers(1 << 20, 64 << 20))
            b[:m].copy_(a[:m], non_blocking=True)
    nhydro.compute_rhs(*resting_column_state(nx, ny, nz))
    n, hist = mg.solve_p(1e-30, 3)
    p, pc = mg.grid(1).p, mg.grid(nl).p
    if ref is None:
        ref, refc, hist0 = p.copy(), pc.copy(), hist.copy()
    elif not (np.array_equal(p, ref) and np.array_equal(pc, refc) and np.array_equal(hist, hist0)):
        bad += 1
        print("rep", rep, "DIFFERS: max |dp| =", float(np.abs(p - ref).max()), flush=True)
    side.synchronize()
solves = nhydro.get_option("coarsest_direct_solves") - d0
mg.nhydro_clean()
out = {"size": [nx, ny, nz], "repetitions": reps, "direct_coarsest_solves": solves, "different_from_first_repetition": bad,
       "p_vs_sweeps": float(np.abs(ref - p_sweeps).max() / np.abs(p_sweeps).max()),
       "history_vs_sweeps": float(np.max(np.abs(hist0[1:] - hist_sweeps[1:]) / hist_sweeps[1:])), "seconds": round(time.time() - t0, 1)}
print(json.dumps(out))
if jpath:
    with open(jpath, "w") as f:
        json.dump(out, f, indent=1)
sys.exit(1 if bad or solves != 18 * reps or out["p_vs_sweeps"] > 1e-12 else 0)
