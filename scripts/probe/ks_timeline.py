#!/usr/bin/env python3
"""Diagnostic: phase timeline of the level-2 colour pass (k_relax_ks<32,8>) from in-kernel stamps.  Needs libmgx.so built with
-DMGX_KS_STAMP (make -C mgroms_amd/csrc EXTRA=-DMGX_KS_STAMP after touching mgx_relax_ks.hip); not part of the product."""
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
for _ in range(5):
    mg.relax(2, 2)
L = lib()
L.mgxk_ks_stamps.argtypes = [C.c_void_p]
buf = np.zeros(1024 * 8, dtype=np.int64)
assert L.mgxk_ks_stamps(buf.ctypes.data) == 0
st = buf.reshape(512, 2, 8)[:256]          # 256 workgroups of the last pass; [wg][wave 0 / wave 7][stamp]
t0 = st[:, :, 0].min()
names = ["entry", "loads issued", "loads done", "rhs parked + barrier", "recurrence + barrier", "stores issued", "stores done"]
clk = 100e6 * (st[:, 0, 6] - st[:, 0, 0]).mean() / 1.0   # placeholder; cycles reported raw
print("stamps in shader-clock cycles relative to the earliest workgroup entry; mean / min / max over 256 workgroups")
for w, wn in ((0, "wave 0"), (1, "wave 7")):
    for q, n in enumerate(names):
        v = st[:, w, q] - t0
        print(f"  {wn} {n:24s} {v.mean():9.0f} {v.min():9.0f} {v.max():9.0f}")
wall = st[:, 0, 7]
print("wall-clock (100 MHz) spread of workgroup entries: %.2f us" % ((wall.max() - wall.min()) / 100.0))
mg.nhydro_clean()
