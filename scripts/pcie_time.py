"""nhydro_solve with HOST arrays (what the Fortran boundary hands over) against the device-resident variant, same work:
the difference is the PCIe traffic of u, v, w (3 x 135 MB each way at 512x512x64)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mgroms_amd as mg
from mgroms_amd import nhydro
from mgroms_amd.testcases import seamount_geometry, resting_column_state

nx, ny, nz = 512, 512, 64
torch.cuda.set_device(0)
nhydro.set_verbose(0)
mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method="FC", solver_maxiter=2, solver_prec=1e-30))
mg.nhydro_matrices(*seamount_geometry(nx, ny), None, 4e3, 0.0, 0.0)
u, v, w = resting_column_state(nx, ny, nz)
ud, vd, wd = (torch.from_numpy(a).cuda() for a in (u, v, w))
up, vp, wp = (torch.from_numpy(a).pin_memory().numpy() for a in (u, v, w))
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
td = t(lambda: nhydro.nhydro_solve_device(ud, vd, wd))
th = t(lambda: mg.nhydro_solve(u, v, w))
tp = t(lambda: mg.nhydro_solve(up, vp, wp))
mb = (u.nbytes + v.nbytes + w.nbytes) * 2 / 1e6
print(f"nhydro_solve (2 F-cycle iterations) 512x512x64: device-resident {td:.2f} ms; host pageable arrays {th:.2f} ms; host pinned arrays {tp:.2f} ms; "
      f"{mb:.0f} MB over PCIe per call -> {mb / (th - td) / 1e3 * 1e3:.1f} GB/s pageable, {mb / (tp - td) / 1e3 * 1e3:.1f} GB/s pinned")
