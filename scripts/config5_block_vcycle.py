import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, mgroms_amd as mg
from mgroms_amd import nhydro
from mgroms_amd.testcases import seamount_geometry, resting_column_state
nx, ny, nz, method = 512, 1024, 128, sys.argv[1]
nhydro.set_verbose(0)
for k, v in (a.split("=") for a in sys.argv[2:]):
    nhydro.set_option(k, int(v))
mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method=method))
mg.nhydro_matrices(*seamount_geometry(nx, ny, 1, 1, 0), None, 4e3, 0.0, 0.0)
nhydro.compute_rhs(*resting_column_state(nx, ny, nz))
nhydro.set_option("async", 1)
for _ in range(2): mg.Vcycle(1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): mg.Vcycle(1)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
nhydro.synchronize(); nhydro.set_option("async", 0)
info = [(nhydro.rbseq_window_info(l), nhydro.rbseq_window_rows(l)) for l in range(1, 3)] if method == "RB" else None
print(method, sys.argv[2:], f"Vcycle(1) {dt:.3f} ms  ({633 * nx * ny * nz / dt / 1e6 / 8000 * 100:.1f} % of 8 TB/s on 633 B/cell)", info)
