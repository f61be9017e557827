#!/usr/bin/env python3
"""level-1 red-black sweep time (sequential order) for a size: python3 scripts/probe_rbseq_sweep.py nx ny nz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mgroms_amd as mg
from mgroms_amd import nhydro
from mgroms_amd.testcases import seamount_geometry, resting_column_state
nx, ny, nz = (int(a) for a in sys.argv[1:4])
torch.cuda.set_device(0); nhydro.set_verbose(0)
mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method="RB"))
mg.nhydro_matrices(*seamount_geometry(nx, ny, 1, 1, 0), None, 4e3, 0.0, 0.0)
nhydro.compute_rhs(*resting_column_state(nx, ny, nz))
nhydro.time_relax(1, 1)
print(nx, ny, nz, "fuse", os.environ.get("MGX_NO_RBSEQ_FUSE", "on"), "poll", os.environ.get("MGX_RBSEQ_POLL"), "sweep_ms", round(min(nhydro.time_relax(1, 20) for _ in range(3)), 4), flush=True)
if os.environ.get("MGX_TRACE"):
    import ctypes
    from mgroms_amd import _lib
    lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libmgx.so"))
    buf = (ctypes.c_ulonglong * 8)()
    lib.mgx_debug_rbs(1, buf)          # clear
    nhydro.time_relax(1, 1)
    lib.mgx_debug_rbs(1, buf)
    t = list(buf); t0 = t[0]
    names = ["walk_start", "last_worker_stores_issued", "last_forward", "last_worker_rows_requested", "last_worker_word_seen", "last_worker_u_read", None, None]
    print({n: (round((v - t0) / 100.0, 2) if v else None) for n, v in zip(names, t) if n}, "us (last colour pass of the sweep)")
mg.nhydro_clean()
