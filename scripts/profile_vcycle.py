#!/usr/bin/env python3
"""Profiling driver: set up the seamount problem and run a few V-cycles (use under rocprofv3 --kernel-trace).
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 scripts/profile_vcycle.py 512 512 64 FC 5"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mgroms_amd as mg  # noqa: E402
from mgroms_amd import nhydro  # noqa: E402
from mgroms_amd.testcases import seamount_geometry  # noqa: E402

nx, ny, nz = (int(a) for a in sys.argv[1:4])
method = sys.argv[4] if len(sys.argv) > 4 else "FC"
ncyc = int(sys.argv[5]) if len(sys.argv) > 5 else 5
torch.cuda.set_device(0)
nhydro.set_verbose(0)
mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method=method))
mg.nhydro_matrices(*seamount_geometry(nx, ny, 1, 1, 0), None, 4e3, 0.0, 0.0)
u = np.zeros((nz, ny + 2, nx + 1)); v = np.zeros((nz, ny + 1, nx + 2)); w = -np.ones((nz + 1, ny + 2, nx + 2)); w[0] = 0
nhydro.compute_rhs(u, v, w)
for _ in range(ncyc):
    mg.Vcycle(1)
print("residual", mg.compute_residual(1))
mg.nhydro_clean()
