#!/usr/bin/env python3
"""Sequential-order red-black: the correction inside the walk's launch (option rbseq_fuse = 1, default) against a launch of its own
behind the walk (0): the same bits (several repetitions: the hand-off between the walk and the correction workers is a cross-XCD
publish), and what it buys on the level-1 sweep / Vcycle / F-cycle iteration.   python3 scripts/rbseq_fuse_check.py [--json path]"""
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

jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
torch.cuda.set_device(0)
nhydro.set_verbose(0)
out = {}
for (nx, ny, nz) in ((512, 512, 64), (256, 256, 32), (128, 256, 16), (64, 128, 8)):
    res = {}
    ref = None
    for fuse, reps in ((0, 1), (1, 6)):
        nhydro.set_option("rbseq_fuse", fuse)
        for rep in range(reps):
            mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method="RB"))
            mg.nhydro_matrices(*seamount_geometry(nx, ny, 1, 1, 0), None, 4e3, 0.0, 0.0)
            nhydro.compute_rhs(*resting_column_state(nx, ny, nz))
            n, hist = mg.solve_p(1e-30, 3)
            p = mg.grid(1).p.copy()
            if ref is None:
                ref = (p, hist.copy())
            else:
                res.setdefault("same_bits", True)
                if not (np.array_equal(p, ref[0]) and np.array_equal(hist, ref[1])):
                    res["same_bits"] = False
                    res["max_diff"] = float(np.abs(p - ref[0]).max())
            if rep == reps - 1:
                nhydro.time_relax(1, 1)
                sweep = min(nhydro.time_relax(1, 20) for _ in range(3))
                mg.Vcycle(1)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20):
                    mg.Vcycle(1)
                torch.cuda.synchronize(); tv = (time.perf_counter() - t0) / 20 * 1e3
                torch.cuda.synchronize(); t0 = time.perf_counter()
                n, _ = mg.solve_p(1e-30, 10)
                torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / n * 1e3
                res["fused" if fuse else "separate"] = {"level1_sweep_ms": round(sweep, 4), "vcycle_ms": round(tv, 4), "fcycle_iteration_ms": round(tf, 4)}
            mg.nhydro_clean()
    out["%dx%dx%d" % (nx, ny, nz)] = res
    print("%dx%dx%d" % (nx, ny, nz), json.dumps(res), flush=True)
nhydro.set_option("rbseq_fuse", 1)
if jpath:
    with open(jpath, "w") as f:
        json.dump(out, f, indent=1)
