#!/usr/bin/env python3
"""Soak test of the cross-XCD hand-off inside the fused red-black launch (k_rbseq_scan FUSE: forwarding waves, sc1 stores / loads, no
acquire; MI355X_MICROARCH.md asks for every word to be checked under uneven load): N repetitions of three solve_p iterations at
512x512x64 with the correction inside the walk's launch, each compared bit for bit with the separate-launch result, while a second
stream of the same process keeps the memory system busy with large device-to-device copies of varying size.
python3 scripts/rbseq_fuse_soak.py [reps [nx ny nz]] [--json path]"""
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
reps = int(args[0]) if args else 60
jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
nx, ny, nz = (int(a) for a in args[1:4]) if len(args) >= 4 else (512, 512, 64)   # e.g. 512 256 8: two workers per plane, workgroups on chunk boundaries
torch.cuda.set_device(0)
nhydro.set_verbose(0)


def run(fuse):
    nhydro.set_option("rbseq_fuse", fuse)
    nhydro.set_option("rbseq_fuse_min", 0)
    mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method="RB"))
    mg.nhydro_matrices(*seamount_geometry(nx, ny, 1, 1, 0), None, 4e3, 0.0, 0.0)
    nhydro.compute_rhs(*resting_column_state(nx, ny, nz))


run(0)
n, hist0 = mg.solve_p(1e-30, 3)
ref = mg.grid(1).p.copy()
mg.nhydro_clean()

run(1)
side = torch.cuda.Stream()
a = torch.empty(64 << 20, dtype=torch.float64, device="cuda")   # 512 MB
b = torch.empty_like(a)
bad, t0 = 0, time.time()
rng = np.random.default_rng(5)
for rep in range(reps):
    with torch.cuda.stream(side):   # uneven background load: copies of 8 .. 512 MB, a random number of them
        for _ in range(int(rng.integers(1, 6))):
            m = int(rng.integers(1 << 20, 64 << 20))
            b[:m].copy_(a[:m], non_blocking=True)
    nhydro.compute_rhs(*resting_column_state(nx, ny, nz))
    n, hist = mg.solve_p(1e-30, 3)
    p = mg.grid(1).p
    if not (np.array_equal(p, ref) and np.array_equal(hist, hist0)):
        bad += 1
        print("rep", rep, "DIFFERS: max |dp| =", float(np.abs(p - ref).max()), flush=True)
    side.synchronize()
still_fused = nhydro.get_option("rbseq_fuse")
mg.nhydro_clean()
nhydro.set_option("rbseq_fuse_min", 4 << 20)
out = {"size": [nx, ny, nz], "repetitions": reps, "fused_launches_per_repetition": "3 F-cycle iterations: 30 level-1 colour passes", "different": bad,
       "rbseq_fuse_still_on": still_fused, "seconds": round(time.time() - t0, 1)}
print(json.dumps(out))
if jpath:
    with open(jpath, "w") as f:
        json.dump(out, f, indent=1)
sys.exit(1 if bad or still_fused != 1 else 0)
