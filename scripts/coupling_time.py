"""Time the model coupling on the GPU: compute_rhs and correct_uvw with u,v,w resident on the device,
plus define_matrices (row f2), at the bench size.  HIP events via torch on the solver's stream."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mgroms_amd as mg
from mgroms_amd import nhydro
from mgroms_amd.testcases import seamount_geometry, resting_column_state

nx, ny, nz = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 512, 64)
torch.cuda.set_device(0)
nhydro.set_verbose(0)
par = nhydro.default_params(relax_method="FC", solver_maxiter=1, solver_prec=1e-30)
mg.nhydro_init(nx, ny, nz, 1, 1, 0, par)
dx, dy, zeta, h = seamount_geometry(nx, ny)
t0 = time.perf_counter(); mg.nhydro_matrices(dx, dy, zeta, h, None, 4e3, 0.0, 0.0); t_first = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(3):
    mg.nhydro_matrices(dx, dy, zeta, h, None, 4e3, 0.0, 0.0)
t_mat = (time.perf_counter() - t0) / 3
u, v, w = (torch.from_numpy(a).cuda() for a in resting_column_state(nx, ny, nz))
nhydro.set_option("tictoc", 1)
for _ in range(5):
    nhydro.nhydro_solve_device(u, v, w)
torch.cuda.synchronize()
nhydro.print_tictoc("/tmp/tictoc_coupling.txt")
print(f"nhydro_matrices: first {t_first*1e3:.1f} ms, repeat {t_mat*1e3:.1f} ms (host wall, includes 4 x {dx.nbytes/1e6:.1f} MB uploads)")
print(open("/tmp/tictoc_coupling.txt").read())
