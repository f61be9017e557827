#!/usr/bin/env python3
"""A model time step as a coupled model with a moving free surface would run it: nhydro_matrices with a new zeta,
then nhydro_solve on device-resident u, v, w (compute_rhs, solve_p, correct_uvw).  Prints wall time per step and per part; under
`rocprofv3 --kernel-trace --stats` the kernels of the three parts.   python3 scripts/profile_timestep.py 512 512 64 [steps] [solver_prec]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mgroms_amd as mg  # noqa: E402
from mgroms_amd import nhydro  # noqa: E402
from mgroms_amd.testcases import seamount_geometry  # noqa: E402

nx, ny, nz = (int(a) for a in sys.argv[1:4])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
prec = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-3
torch.cuda.set_device(0)
nhydro.set_verbose(0)
mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method="FC", solver_prec=prec))
dx, dy, zeta, h = seamount_geometry(nx, ny, 1, 1, 0)
dev = torch.device("cuda:0")
u = torch.zeros((nz, ny + 2, nx + 1), dtype=torch.float64, device=dev)
v = torch.zeros((nz, ny + 1, nx + 2), dtype=torch.float64, device=dev)
w = -torch.ones((nz + 1, ny + 2, nx + 2), dtype=torch.float64, device=dev)
w[0] = 0
jj, ii = np.meshgrid(np.arange(ny + 2), np.arange(nx + 2))  # arrays are (0:nx+1, 0:ny+1) with j fastest
t_mat = t_sol = 0.0
for s in range(steps + 1):
    zeta_s = 0.05 * np.sin(2 * np.pi * (ii / nx + 0.1 * s)) * np.cos(2 * np.pi * jj / ny)  # a free surface that moves
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mg.nhydro_matrices(dx, dy, np.ascontiguousarray(zeta_s), h, None, 4e3, 0.0, 0.0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    nhydro.nhydro_solve_device(u, v, w)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if s:  # the first step pays the one-off allocations
        t_mat += t1 - t0
        t_sol += t2 - t1
print("per time step: nhydro_matrices %.2f ms (host arrays in, %d levels rebuilt), nhydro_solve on resident u,v,w %.2f ms" % (t_mat / steps * 1e3, mg.nlevs(), t_sol / steps * 1e3))
mg.nhydro_clean()
