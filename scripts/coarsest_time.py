"""Time relax(nlevs, ns_coarsest) -- the coarsest-level solve of every V-cycle -- on the 512x512x64 hierarchy (16x16x2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mgroms_amd as mg
from mgroms_amd import nhydro
from mgroms_amd.testcases import seamount_geometry
method = sys.argv[1] if len(sys.argv) > 1 else "FC"
nx, ny, nz = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (512, 512, 64)
torch.cuda.set_device(0); nhydro.set_verbose(0)
mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method=method))
mg.nhydro_matrices(*seamount_geometry(nx, ny), None, 4e3, 0.0, 0.0)
L = mg.nlevs()
g = mg.grid(L)
for _ in range(20): mg.relax(L, 40)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 300
for _ in range(n): mg.relax(L, 40)
torch.cuda.synchronize()
print(f"{method} level {L} {g.nx}x{g.ny}x{g.nz}: relax(40 sweeps) {(time.perf_counter()-t0)/n*1e6:.1f} us per call (incl. launch + sync)")
