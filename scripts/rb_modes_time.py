#!/usr/bin/env python3
"""Red-black (relax_method='RB', cmatrix='real': the reference default) in its three modes -- plain parallel sweep (rb_seq = 0), the
reference's sequential order at speed (rb_seq = 1, default) and the same order bit for bit with one launch per plane (rb_exact = 1):
level-1 sweep (HIP events), Vcycle(1) and F-cycle iteration rates, and how far each mode's iterate is from the exact order's.
python3 scripts/rb_modes_time.py [nx ny nz] [--exact-reps N] [--json path]"""
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
nx, ny, nz = (int(a) for a in args[:3]) if len(args) >= 3 else (512, 512, 64)
jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
torch.cuda.set_device(0)
nhydro.set_verbose(0)
out = {"size": [nx, ny, nz], "modes": {}}
fields = {}
for mode, opts in (("exact", {"rb_exact": 1, "rb_seq": 0}), ("parallel", {"rb_exact": 0, "rb_seq": 0}),
                   ("sequential_walk", {"rb_exact": 0, "rb_seq": 1, "rbseq_window": 0}),      # the walk over the whole level (round 4's first form)
                   ("sequential", {"rb_exact": 0, "rb_seq": 1, "rbseq_window": 1})):          # the windowed walk (default)
    for k, v in opts.items():
        nhydro.set_option(k, v)
    mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method="RB"))
    mg.nhydro_matrices(*seamount_geometry(nx, ny, 1, 1, 0), None, 4e3, 0.0, 0.0)
    nhydro.compute_rhs(*resting_column_state(nx, ny, nz))
    if mode == "sequential":
        out["window"] = {f"level{l}": dict(zip(("rho", "planes"), nhydro.rbseq_window_info(l))) for l in range(1, mg.nlevs() + 1)}
    n, hist = mg.solve_p(1e-30, 2)          # two iterations from p = 0: the iterate the modes are compared on
    fields[mode] = (mg.grid(1).p, hist.copy())
    reps = 2 if mode == "exact" else 20
    nhydro.time_relax(1, 1)
    sweep = min(nhydro.time_relax(1, reps) for _ in range(2 if mode == "exact" else 3))
    nv = 2 if mode == "exact" else 20
    mg.Vcycle(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(nv):
        mg.Vcycle(1)
    torch.cuda.synchronize(); tv = (time.perf_counter() - t0) / nv * 1e3
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n, _ = mg.solve_p(1e-30, 1 if mode == "exact" else 10)
    torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / n * 1e3
    out["modes"][mode] = {"level1_sweep_ms": round(sweep, 4), "vcycle_ms": round(tv, 4), "fcycle_iteration_ms": round(tf, 4)}
    if mode.startswith("sequential"):
        out["modes"][mode]["window_colours"] = nhydro.get_option("rbseq_window_colours")
    mg.nhydro_clean()
nhydro.set_option("rb_exact", 0); nhydro.set_option("rb_seq", 1); nhydro.set_option("rbseq_window", 1)
pe, he = fields["exact"]
for mode in ("parallel", "sequential_walk", "sequential"):
    p, h = fields[mode]
    out["modes"][mode]["p_vs_exact_order"] = float(np.abs(p - pe).max() / np.abs(pe).max())
    out["modes"][mode]["residual_vs_exact_order"] = float(np.max(np.abs(h[1:] - he[1:]) / he[1:]))
m = out["modes"]
out["sweep_ratio_sequential_over_parallel"] = round(m["sequential"]["level1_sweep_ms"] / m["parallel"]["level1_sweep_ms"], 3)
out["vcycle_ratio_sequential_over_parallel"] = round(m["sequential"]["vcycle_ms"] / m["parallel"]["vcycle_ms"], 3)
print(json.dumps(out, indent=1))
if jpath:
    with open(jpath, "w") as f:
        json.dump(out, f, indent=1)
