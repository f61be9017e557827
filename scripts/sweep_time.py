#!/usr/bin/env python3
"""Time level-1 smoother sweeps and residuals only (HIP events): python3 scripts/sweep_time.py [nx ny nz method reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mgroms_amd as mg  # noqa: E402
from mgroms_amd import nhydro  # noqa: E402
from mgroms_amd.testcases import seamount_geometry  # noqa: E402

nx, ny, nz = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 512, 64)
method = sys.argv[4] if len(sys.argv) > 4 else "FC"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
torch.cuda.set_device(0)
nhydro.set_verbose(0)
mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method=method))
mg.nhydro_matrices(*seamount_geometry(nx, ny, 1, 1, 0), None, 4e3, 0.0, 0.0)
u = np.zeros((nz, ny + 2, nx + 1)); v = np.zeros((nz, ny + 1, nx + 2)); w = -np.ones((nz + 1, ny + 2, nx + 2)); w[0] = 0
nhydro.compute_rhs(u, v, w)
mg.Vcycle(1)
nhydro.time_relax(1, 5)
s = [nhydro.time_relax(1, reps) for _ in range(3)]
r = [nhydro.time_residual(1, reps) for _ in range(3)]
cells = nx * ny * nz
print(f"{method} {nx}x{ny}x{nz} D={os.environ.get('MGX_D','-')}: sweep {min(s):.4f} ms ({88*cells/min(s)/1e6/8000*100:.1f}% of 8 TB/s), residual {min(r):.4f} ms ({88*cells/min(r)/1e6/8000*100:.1f}%)")
mg.nhydro_clean()
