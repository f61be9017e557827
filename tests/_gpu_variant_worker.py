"""Worker of test_transfer_kernel_variants_same_bits: the transfer-kernel switches are read from the environment once per process, so every
variant runs in a process of its own and prints a digest of every level's p after two V-cycles and two F-cycle iterations."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mgroms_amd as mg  # noqa: E402
from mgroms_amd import nhydro  # noqa: E402
from mgroms_amd.testcases import seamount_geometry  # noqa: E402

nx, ny, nz = (int(a) for a in sys.argv[1:4])
torch.cuda.set_device(0)
nhydro.set_verbose(0)
mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method="FC"))
mg.nhydro_matrices(*seamount_geometry(nx, ny, 1, 1, 0), None, 4e3, 0.0, 0.0)
rng = np.random.default_rng(5)
u = 1e-2 * rng.standard_normal((nz, ny + 2, nx + 1)); v = 1e-2 * rng.standard_normal((nz, ny + 1, nx + 2)); w = -np.ones((nz + 1, ny + 2, nx + 2)); w[0] = 0
nhydro.compute_rhs(u, v, w)
h = hashlib.sha256()
h.update(mg.grid(1).b.tobytes())
for _ in range(2):
    mg.Vcycle(1)
for lev in range(1, mg.nlevs() + 1):
    h.update(mg.grid(lev).p.tobytes()); h.update(mg.grid(lev).b.tobytes())
n, hist = mg.solve_p(1e-30, 2)
for lev in range(1, mg.nlevs() + 1):
    h.update(mg.grid(lev).p.tobytes())
h.update(np.asarray(hist).tobytes())
nhydro.nhydro_solve(u, v, w)
h.update(u.tobytes()); h.update(v.tobytes()); h.update(w.tobytes())
print("DIGEST", h.hexdigest())
mg.nhydro_clean()
