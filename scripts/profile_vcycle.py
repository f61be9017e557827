#!/usr/bin/env python3
"""Profiling driver: N Vcycle(1) on the seamount problem (after one solve_p iteration, so that every level holds a right-hand side).
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/vc -- python3 scripts/profile_vcycle.py 512 512 64 RB 10"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mgroms_amd as mg  # noqa: E402
from mgroms_amd import nhydro  # noqa: E402
from mgroms_amd.testcases import seamount_geometry, resting_column_state  # noqa: E402

nx, ny, nz = (int(a) for a in sys.argv[1:4])
method = sys.argv[4] if len(sys.argv) > 4 else "FC"
nit = int(sys.argv[5]) if len(sys.argv) > 5 else 10
torch.cuda.set_device(0)
nhydro.set_verbose(0)
mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method=method))
mg.nhydro_matrices(*seamount_geometry(nx, ny, 1, 1, 0), None, 4e3, 0.0, 0.0)
nhydro.compute_rhs(*resting_column_state(nx, ny, nz))
mg.solve_p(1e-30, 1)
for _ in range(nit):
    mg.Vcycle(1)
torch.cuda.synchronize()
mg.nhydro_clean()
