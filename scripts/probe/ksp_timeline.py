#!/usr/bin/env python3
"""Diagnostic: phase timeline of the persistent mid-level relax (k_relax_ksp) on level 3 of the 512x512x64 hierarchy, second sweep of a
three-sweep call, from in-kernel stamps of the 100 MHz constant clock (10 ns resolution, the same clock on every CU).  Needs libmgx.so built
with -DMGX_KS_STAMP; not part of the product."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import mgroms_amd as mg  # noqa: E402
from mgroms_amd import nhydro  # noqa: E402
from mgroms_amd._lib import lib  # noqa: E402
from mgroms_amd.testcases import seamount_geometry, resting_column_state  # noqa: E402

torch.cuda.set_device(0)
nhydro.set_verbose(0)
mg.nhydro_init(512, 512, 64, 1, 1, 0, nhydro.default_params(relax_method="FC"))
mg.nhydro_matrices(*seamount_geometry(512, 512), None, 4e3, 0.0, 0.0)
nhydro.compute_rhs(*resting_column_state(512, 512, 64))
mg.Vcycle(1)
lev = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(3):
    mg.relax(lev, 3)
L = lib()
L.mgxk_ks_stamps.argtypes = [C.c_void_p]
buf = np.zeros(1024 * 8, dtype=np.int64)
assert L.mgxk_ks_stamps(buf.ctypes.data) == 0
nx = mg.grid(lev).nx
st = buf.reshape(1024, 8)[:nx].astype(np.float64) * 0.01   # us
names = ["sweep start", "neighbours seen (+acquire)", "colour a: loads + rhs parked", "colour a recurrence", "colour b rhs + recurrence", "stores drained", "published"]
t0 = st[:, 0].min()
print(f"level {lev}: {nx} planes, second sweep; microseconds; per plane relative to ITS sweep start: mean (min..max)")
for q in range(1, 7):
    d = st[:, q] - st[:, 0]
    print(f"  {names[q]:34s} {d.mean():6.2f} ({d.min():5.2f} .. {d.max():5.2f})")
odd = st[0::2]; even = st[1::2]   # blockIdx order is not plane order; parity statistics are indicative only
print("sweep start spread over planes: %.2f us; whole second sweep (first start to last publish): %.2f us" % (st[:, 0].max() - t0, st[:, 6].max() - t0))
mg.nhydro_clean()
