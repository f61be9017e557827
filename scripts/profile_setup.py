#!/usr/bin/env python3
"""Profiling driver: nhydro_matrices only (define_matrices on the device), n rebuilds.
  rocprofv3 --kernel-trace --stats ... -- python3 scripts/profile_setup.py 512 512 64 3"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mgroms_amd as mg  # noqa: E402
from mgroms_amd import nhydro  # noqa: E402
from mgroms_amd.testcases import seamount_geometry  # noqa: E402

nx, ny, nz = (int(a) for a in sys.argv[1:4])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 3
torch.cuda.set_device(0)
nhydro.set_verbose(0)
mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method=(sys.argv[5] if len(sys.argv) > 5 else "FC")))
geo = seamount_geometry(nx, ny, 1, 1, 0)
mg.nhydro_matrices(*geo, None, 4e3, 0.0, 0.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    mg.nhydro_matrices(*geo, None, 4e3, 0.0, 0.0)
torch.cuda.synchronize()
print(f"nhydro_matrices {nx}x{ny}x{nz}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per rebuild (all levels, host arrays in)")
mg.nhydro_clean()
