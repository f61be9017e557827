"""Host-side mirror of the reference's model-facing API (src/nhydro.f90) and of the solver-level entry points
the reference's tests drive (src/mg_solvers.f90, mg_relax.f90, mg_intergrids.f90), same names and argument
meaning.  Arrays follow the reference's Fortran shapes; pass numpy arrays whose C-order index is the reversed
Fortran index, i.e. a field p(nz,0:ny+1,0:nx+1) is a numpy array of shape (nx+2, ny+2, nz).
"""
import ctypes as C
import threading

import numpy as np

from ._lib import Params, MgxError, lib, check  # noqa: F401

FIELD = {"p": 0, "b": 1, "r": 2, "cA": 3, "dx": 4, "dy": 5, "zeta": 6, "h": 7, "zr": 8, "zw": 9, "cw": 10, "rmask": 14}
_DP = C.POINTER(C.c_double)


class _PerThread(threading.local):
    """dims of the instance the calling thread drives (include/mgx.h: the instance selection is per thread as well)"""
    dims = None


_state = _PerThread()


def _dp(a):
    return a.ctypes.data_as(_DP)


def _f64(a, shape=None, name="array"):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


def default_params(**kw):
    p = Params()
    check(lib().mgx_params_default(C.byref(p)))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(f"{k} is not a member of namelist /nhparam/")
        setattr(p, k, v.encode() if isinstance(v, str) else v)
    return p


def read_nhnamelist(filename="nh_namelist", **overrides):
    """read_nhnamelist (mg_namelist.f90:55): defaults, then the file if it exists, then keyword overrides."""
    p = default_params()
    check(lib().mgx_read_namelist(filename.encode(), C.byref(p)))
    for k, v in overrides.items():
        setattr(p, k, v.encode() if isinstance(v, str) else v)
    return p


def set_option(name, value):
    """warm_start / tictoc / exact_halos / verbose (include/mgx.h: mgx_set_option)."""
    check(lib().mgx_set_option(name.encode(), int(value)))


def synchronize():
    """wait for the solver's stream and raise what the device flagged (counterpart of set_option("async", 1))"""
    check(lib().mgx_synchronize())


def get_option(name):
    """read back an option or an integer / logical member of /nhparam/ (include/mgx.h: mgx_get_option)"""
    v = C.c_int()
    check(lib().mgx_get_option(name.encode(), C.byref(v)))
    return v.value


def print_tictoc(path="fort.10"):
    """print_tictoc (mg_tictoc.f90:114): per-level timer table of relax / residual / Fcycle / solve / compute_rhs."""
    check(lib().mgx_print_tictoc(path.encode()))


def set_verbose(level):
    check(lib().mgx_set_verbose(int(level)))


def nhydro_init(nx, ny, nz, npxg=1, npyg=1, rank=0, params=None, comm=None):
    """nhydro_init(nx,ny,nz,npxg,npyg) (nhydro.f90:18).  `params=None` reads ./nh_namelist like the reference.
    `comm`: an mgroms_amd.parallel.Comm when npxg*npyg > 1 (one process per GPU)."""
    if comm is not None:
        comm.install()
    check(lib().mgx_init(nx, ny, nz, npxg, npyg, rank, None if params is None else C.byref(params)))
    _state.dims = (nx, ny, nz)
    if comm is not None and npxg * npyg > 1:
        comm.after_init()


def nhydro_matrices(dx, dy, zeta, h, rmask=None, hc=0.0, theta_b=0.0, theta_s=0.0):
    """nhydro_matrices (nhydro.f90:36): 2-D arrays are (0:ny+1,0:nx+1) in Fortran = numpy shape (nx+2, ny+2)."""
    nx, ny, _ = _state.dims
    sh = (nx + 2, ny + 2)
    dx, dy, zeta, h = (_f64(a, sh, n) for a, n in ((dx, "dx"), (dy, "dy"), (zeta, "zeta"), (h, "h")))
    rm = None if rmask is None else _f64(rmask, sh, "rmask")
    check(lib().mgx_matrices(_dp(dx), _dp(dy), _dp(zeta), _dp(h), None if rm is None else _dp(rm), hc, theta_b, theta_s))


def _uvw(u, v, w):
    nx, ny, nz = _state.dims
    for a, sh, n in ((u, (nz, ny + 2, nx + 1), "u"), (v, (nz, ny + 1, nx + 2), "v"), (w, (nz + 1, ny + 2, nx + 2), "w")):
        if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.shape == sh):
            raise ValueError(f"{n}: need a C-contiguous float64 array of shape {sh} (updated in place)")


def _mask(rmask):
    """rmaska of the call, (0:ny+1,0:nx+1) in Fortran = numpy (nx+2, ny+2); None = the mask of nhydro_matrices / all ones."""
    if rmask is None:
        return None, None
    nx, ny, _ = _state.dims
    rm = _f64(rmask, (nx + 2, ny + 2), "rmask")
    return rm, _dp(rm)


def nhydro_solve(u, v, w, rmask=None):
    """nhydro_solve (nhydro.f90:53): u(1:nx+1,0:ny+1,1:nz) = numpy (nz, ny+2, nx+1) etc.; corrected in place."""
    _uvw(u, v, w)
    rm, prm = _mask(rmask)
    check(lib().mgx_solve(_dp(u), _dp(v), _dp(w), prm))


def nhydro_solve_device(u, v, w, rmask=None):
    """nhydro_solve on torch CUDA tensors (float64, contiguous, shapes as nhydro_solve; rmask (nx+2, ny+2)): no host round trip."""
    nx, ny, nz = _state.dims
    for a, sh, n in ((u, (nz, ny + 2, nx + 1), "u"), (v, (nz, ny + 1, nx + 2), "v"), (w, (nz + 1, ny + 2, nx + 2), "w")):
        if not (a.is_cuda and a.is_contiguous() and tuple(a.shape) == sh and str(a.dtype) == "torch.float64"):
            raise ValueError(f"{n}: need a contiguous float64 CUDA tensor of shape {sh}")
    import torch
    torch.cuda.current_stream().synchronize()
    if rmask is not None and not (rmask.is_cuda and rmask.is_contiguous() and tuple(rmask.shape) == (nx + 2, ny + 2) and str(rmask.dtype) == "torch.float64"):
        raise ValueError(f"rmask: need a contiguous float64 CUDA tensor of shape {(nx + 2, ny + 2)}")
    check(lib().mgx_solve_device(C.c_void_p(u.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(w.data_ptr()),
                                 None if rmask is None else C.c_void_p(rmask.data_ptr())))


def nhydro_check_nondivergence(u, v, w, rmask=None):
    _uvw(u, v, w)
    rm, prm = _mask(rmask)
    check(lib().mgx_check_nondivergence(_dp(u), _dp(v), _dp(w), prm))


def compute_rhs(u, v, w, rmask=None):
    _uvw(u, v, w)
    rm, prm = _mask(rmask)
    check(lib().mgx_compute_rhs(_dp(u), _dp(v), _dp(w), prm))


def nhydro_clean():
    lib().mgx_clean()
    _state.dims = None


# ---- mg_solvers / mg_relax / mg_intergrids ------------------------------------------------------
def solve_p(tol, maxite):
    """solve_p(tol,maxite) (mg_solvers.f90:17).  Returns (nite, history) with history[0] the initial ||r||/||b||."""
    n = C.c_int()
    res = C.c_double()
    hist = (C.c_double * (maxite + 1))()
    check(lib().mgx_solve_p(tol, maxite, C.byref(n), C.byref(res), hist))
    return n.value, np.array(hist[:n.value + 1])


def Fcycle():
    check(lib().mgx_fcycle())


def Vcycle(lev=1):
    check(lib().mgx_vcycle(lev))


def Vcycle2(lev1, lev2):
    check(lib().mgx_vcycle2(lev1, lev2))


def relax(lev, nsweeps):
    check(lib().mgx_relax(lev, nsweeps))


def compute_residual(lev):
    """compute_residual(lev,res) (mg_relax.f90:337): returns the global L2 norm of r."""
    r = C.c_double()
    check(lib().mgx_residual(lev, C.byref(r)))
    return r.value


def fine2coarse(lev):
    check(lib().mgx_fine2coarse(lev))


def coarse2fine(lev):
    check(lib().mgx_coarse2fine(lev))


def fill_halo(lev, name):
    check(lib().mgx_fill_halo(lev, FIELD[name]))


def testgalerkin(lev):
    """testgalerkin(lev) (mg_solvers.f90:203): returns (norm_c, norm_f) for the coarse field found in grid(lev).p."""
    a, b = C.c_double(), C.c_double()
    check(lib().mgx_testgalerkin(lev, C.byref(a), C.byref(b)))
    return a.value, b.value


def nlevs():
    return lib().mgx_nlevs()


def rbseq_window_info(lev):
    """(rho, planes): the contraction bound of the level's red-black walk and the planes of warm-up of the windowed walk
    (option "rbseq_window", include/mgx.h); planes = 0: the walk over the whole level stays."""
    rho, m = C.c_double(), C.c_int()
    check(lib().mgx_rbseq_window_info(lev, C.byref(rho), C.byref(m)))
    return rho.value, m.value


def rbseq_window_rows(lev):
    """rows (from the bottom) the windowed walk's correction reaches on the level (option "rbseq_rowcut"); nz = every row."""
    r = C.c_int()
    check(lib().mgx_rbseq_window_rows(lev, C.byref(r)))
    return r.value


class _Level:
    """grid(lev) (mg_grids.f90:24-65): dims, decomposition info and host copies of the level's arrays."""

    def __init__(self, lev):
        self.lev = lev
        nx, ny, nz = C.c_int(), C.c_int(), C.c_int()
        check(lib().mgx_level_dims(lev, C.byref(nx), C.byref(ny), C.byref(nz)))
        self.nx, self.ny, self.nz = nx.value, ny.value, nz.value
        info = (C.c_int * 18)()
        check(lib().mgx_level_info(lev, info))
        (self.npx, self.npy, self.incx, self.incy, self.gather, self.ngx, self.ngy, self.key, self.color) = list(info[:9])
        self.neighb = list(info[10:18])

    def _shape(self, name):
        nx, ny, nz = self.nx, self.ny, self.nz
        return {"p": (nx + 2, ny + 2, nz), "b": (nx + 2, ny + 2, nz), "r": (nx + 2, ny + 2, nz),
                "cA": (nx + 2, ny + 2, nz, 8), "dx": (nx + 2, ny + 2), "dy": (nx + 2, ny + 2),
                "zeta": (nx + 2, ny + 2), "h": (nx + 2, ny + 2), "rmask": (nx + 2, ny + 2), "zr": (nx + 4, ny + 4, nz),
                "zw": (nx + 4, ny + 4, nz + 1), "cw": (nx + 2, ny + 2, nz + 1)}[name]

    def get(self, name):
        a = np.empty(self._shape(name), dtype=np.float64)
        check(lib().mgx_get_field(self.lev, FIELD[name], _dp(a)))
        return a

    def set(self, name, a):
        a = _f64(a, self._shape(name), name)
        check(lib().mgx_set_field(self.lev, FIELD[name], _dp(a)))

    def __getattr__(self, name):
        if name in FIELD:
            return self.get(name)
        raise AttributeError(name)


def grid(lev):
    return _Level(lev)


def level_table(nx, ny, nz, npx=1, npy=1, rank=0, nsmall=8):
    """Level hierarchy of `rank` (mg_grids.f90:468-738) as a list of dicts; pure host logic, no GPU needed."""
    out = (C.c_int * (20 * 32))()
    nl = lib().mgx_level_table(nx, ny, nz, npx, npy, rank, nsmall, 32, out)
    if nl < 0:
        raise MgxError("level_table: invalid arguments")
    keys = ["nx", "ny", "nz", "npx", "npy", "incx", "incy", "gather", "ngx", "ngy", "key", "color"]
    res = []
    for l in range(nl):
        d = dict(zip(keys, list(out[20 * l:20 * l + 12])))
        d["neighb"] = list(out[20 * l + 12:20 * l + 20])
        res.append(d)
    return res


# ---- measurement helpers (bench.py) -------------------------------------------------------------
def time_relax(lev, reps):
    ms = C.c_float()
    check(lib().mgx_time_relax(lev, reps, C.byref(ms)))
    return ms.value


def time_residual(lev, reps):
    ms = C.c_float()
    check(lib().mgx_time_residual(lev, reps, C.byref(ms)))
    return ms.value


def counters():
    out = (C.c_longlong * 4)()
    check(lib().mgx_counters(out))
    d = dict(zip(("launches", "halo_fills", "exchanges", "allreduces"), list(out)))
    d["p2p_exchanges"] = int(lib().mgx_p2p_exchanges())
    return d
