"""mgroms_amd -- MI355X-native multigrid pressure solve behind the reference's nhydro / mg_solvers interface.

The package is a thin host mirror of the reference's Fortran entry points over the C ABI of libmgx.so
(include/mgx.h).  All computation happens in hand-written HIP kernels (mgroms_amd/csrc); there is no CPU
fallback: importing works anywhere, but every solver call needs the built library and a GPU.
"""
from . import nhydro  # noqa: F401
from .nhydro import (nhydro_init, nhydro_matrices, nhydro_solve, nhydro_check_nondivergence, nhydro_clean,  # noqa: F401
                     solve_p, Fcycle, Vcycle, Vcycle2, relax, compute_residual, fine2coarse, coarse2fine, fill_halo,
                     grid, nlevs, Params, read_nhnamelist)

__all__ = ["nhydro", "nhydro_init", "nhydro_matrices", "nhydro_solve", "nhydro_check_nondivergence", "nhydro_clean",
           "solve_p", "Fcycle", "Vcycle", "Vcycle2", "relax", "compute_residual", "fine2coarse", "coarse2fine", "fill_halo",
           "grid", "nlevs", "Params", "read_nhnamelist"]
