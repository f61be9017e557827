#!/usr/bin/env python3
"""Time-to-solution of solve_p on the seamount problem: python3 scripts/solve_time.py nx ny nz method tol maxite"""
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
method, tol, maxite = sys.argv[4], float(sys.argv[5]), int(sys.argv[6])
torch.cuda.set_device(0)
nhydro.set_verbose(0)
t0 = time.perf_counter()
mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method=method, solver_prec=tol, solver_maxiter=maxite))
mg.nhydro_matrices(*seamount_geometry(nx, ny, 1, 1, 0), None, 4e3, 0.0, 0.0)
torch.cuda.synchronize()
t1 = time.perf_counter()
u = np.zeros((nz, ny + 2, nx + 1)); v = np.zeros((nz, ny + 1, nx + 2)); w = -np.ones((nz + 1, ny + 2, nx + 2)); w[0] = 0
nhydro.compute_rhs(u, v, w)
mg.solve_p(tol, 2)  # warm-up
torch.cuda.synchronize()
t2 = time.perf_counter()
n, hist = mg.solve_p(tol, maxite)
torch.cuda.synchronize()
t3 = time.perf_counter()
print(f"{method} {nx}x{ny}x{nz} tol={tol:g}: init+matrices {t1-t0:.3f} s; solve_p {n} iterations in {(t3-t2)*1e3:.1f} ms "
      f"({(t3-t2)/max(n,1)*1e3:.2f} ms/it), res {hist[-1]:.3e}, first5 {[float('%.3g' % h) for h in hist[1:6]]}")
mg.nhydro_clean()
