#!/usr/bin/env python3
"""Vcycle / solve_p iteration time of the sequential-order red-black for several values of option rbseq_fuse_min: python3 scripts/probe_fuse_min.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mgroms_amd as mg
from mgroms_amd import nhydro
from mgroms_amd.testcases import seamount_geometry, resting_column_state
nx, ny, nz = 512, 512, 64
torch.cuda.set_device(0); nhydro.set_verbose(0)
for fm in (4 << 20, 1 << 20, 4 << 20, 1 << 20):
    nhydro.set_option("rbseq_fuse_min", fm)
    mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method="RB"))
    mg.nhydro_matrices(*seamount_geometry(nx, ny, 1, 1, 0), None, 4e3, 0.0, 0.0)
    nhydro.compute_rhs(*resting_column_state(nx, ny, nz))
    mg.solve_p(1e-30, 2)
    mg.Vcycle(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        mg.Vcycle(1)
    torch.cuda.synchronize(); tv = (time.perf_counter() - t0) / 20 * 1e3
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n, _ = mg.solve_p(1e-30, 10)
    torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / n * 1e3
    print("rbseq_fuse_min", fm, "vcycle_ms", round(tv, 4), "fcycle_iteration_ms", round(tf, 4), flush=True)
    mg.nhydro_clean()
nhydro.set_option("rbseq_fuse_min", 4 << 20)
