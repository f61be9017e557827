"""ctypes front-end of the CPU oracle (oracle/mgoracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (mgroms_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

# The emulated ranks are spread over OpenMP threads.  On a many-core host with a small CPU quota (the GPU boxes: 16 cores'
# worth of a much larger machine) libgomp's default -- one spinning thread per visible CPU -- starves the one thread that
# has work: cap the team unless the caller chose otherwise.  Must happen before libgomp loads.
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 16)))

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmgoracle.so")

FIELD = {"p": 0, "b": 1, "r": 2, "cA": 3, "dx": 4, "dy": 5, "zeta": 6, "h": 7,
         "zr": 8, "zw": 9, "cw": 10, "u": 11, "v": 12, "w": 13, "rmask": 14, "rmaska": 15}
METHOD = {"GS": 0, "Gauss-Seidel": 0, "RB": 1, "Red-Black": 1, "FC": 2, "Four-Color": 2}


class Params(C.Structure):
    """The &nhparam members the path reads (mg_namelist.f90:11-35)."""
    _fields_ = [("solver_prec", C.c_double), ("solver_maxiter", C.c_int), ("nsmall", C.c_int),
                ("ns_coarsest", C.c_int), ("ns_pre", C.c_int), ("ns_post", C.c_int),
                ("cmatrix_real", C.c_int), ("relax_method", C.c_int), ("interp_linear", C.c_int), ("bmask", C.c_int)]


def build(force=False):
    src = os.path.join(_HERE, "mgoracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libmgoracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.mgo_create.restype = C.c_void_p
        L.mgo_create.argtypes = [C.c_int] * 5 + [C.POINTER(Params)]
        L.mgo_destroy.argtypes = [C.c_void_p]
        L.mgo_nlevs.argtypes = [C.c_void_p]
        L.mgo_level_info.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.mgo_field.restype = C.POINTER(C.c_double)
        L.mgo_field.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.mgo_matrices.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        for n in ("mgo_compute_rhs", "mgo_correct_uvw", "mgo_fcycle", "mgo_check_nondivergence"):
            getattr(L, n).argtypes = [C.c_void_p]
        L.mgo_solve_p.restype = C.c_int
        L.mgo_solve_p.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.mgo_nhydro_solve.restype = C.c_int
        L.mgo_nhydro_solve.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.mgo_vcycle.argtypes = [C.c_void_p, C.c_int]
        L.mgo_relax.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mgo_residual.restype = C.c_double
        L.mgo_residual.argtypes = [C.c_void_p, C.c_int]
        L.mgo_fine2coarse.argtypes = [C.c_void_p, C.c_int]
        L.mgo_coarse2fine.argtypes = [C.c_void_p, C.c_int]
        L.mgo_fill_halo.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mgo_use_call_mask.argtypes = [C.c_void_p, C.c_int]
        _lib = L
    return _lib


class Oracle:
    """One emulated MPI world of npx*npy ranks, local block nx x ny x nz each."""

    def __init__(self, nx, ny, nz, npx=1, npy=1, relax_method="RB", solver_prec=1e-6, solver_maxiter=50,
                 nsmall=8, ns_coarsest=40, ns_pre=3, ns_post=2, cmatrix="real", interp_type="linear", bmask=False):
        self.par = Params(solver_prec, solver_maxiter, nsmall, ns_coarsest, ns_pre, ns_post,
                          1 if cmatrix == "real" else 0, METHOD[relax_method], 1 if interp_type == "linear" else 0,
                          1 if bmask else 0)
        self.nx, self.ny, self.nz, self.npx, self.npy = nx, ny, nz, npx, npy
        self.nranks = npx * npy
        self.h = lib().mgo_create(nx, ny, nz, npx, npy, C.byref(self.par))
        self.nlevs = lib().mgo_nlevs(self.h)

    def close(self):
        if self.h:
            lib().mgo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def level_info(self, lev, rank=0):
        out = (C.c_int * 20)()
        lib().mgo_level_info(self.h, rank, lev, out)
        keys = ["nx", "ny", "nz", "npx", "npy", "incx", "incy", "gather", "ngx", "ngy", "key", "color"]
        d = dict(zip(keys, list(out[:12])))
        d["neighb"] = list(out[12:20])
        return d

    def field(self, name, lev=1, rank=0):
        """numpy VIEW of an oracle array, C-order index [i][j][k] (= Fortran (k,j,i))."""
        li = self.level_info(lev, rank)
        nx, ny, nz = li["nx"], li["ny"], li["nz"]
        l1 = self.level_info(1, rank)
        shape = {"p": (nx + 2, ny + 2, nz), "b": (nx + 2, ny + 2, nz), "r": (nx + 2, ny + 2, nz),
                 "cA": (nx + 2, ny + 2, nz, 8), "dx": (nx + 2, ny + 2), "dy": (nx + 2, ny + 2),
                 "zeta": (nx + 2, ny + 2), "h": (nx + 2, ny + 2), "rmask": (nx + 2, ny + 2), "zr": (nx + 4, ny + 4, nz),
                 "zw": (nx + 4, ny + 4, nz + 1), "cw": (nx + 2, ny + 2, nz + 1), "rmaska": (l1["nx"] + 2, l1["ny"] + 2),
                 "u": (l1["nz"], l1["ny"] + 2, l1["nx"] + 1), "v": (l1["nz"], l1["ny"] + 1, l1["nx"] + 2),
                 "w": (l1["nz"] + 1, l1["ny"] + 2, l1["nx"] + 2)}[name]
        ptr = lib().mgo_field(self.h, rank, lev, FIELD[name])
        return np.ctypeslib.as_array(ptr, shape=shape)

    # -- the reference's entry points -------------------------------------
    def matrices(self, hc, theta_b, theta_s):
        lib().mgo_matrices(self.h, hc, theta_b, theta_s)

    def compute_rhs(self):
        lib().mgo_compute_rhs(self.h)

    def correct_uvw(self):
        lib().mgo_correct_uvw(self.h)

    def use_call_mask(self, on=True):
        """compute_rhs / correct_uvw read field "rmaska" (the mask handed to nhydro_solve, nhydro.f90:56,72) instead of
        the level-1 mask of nhydro_matrices."""
        lib().mgo_use_call_mask(self.h, 1 if on else 0)

    def solve_p(self, tol=None, maxite=None):
        tol = self.par.solver_prec if tol is None else tol
        maxite = self.par.solver_maxiter if maxite is None else maxite
        hist = (C.c_double * (maxite + 1))()
        bn = C.c_double()
        n = lib().mgo_solve_p(self.h, tol, maxite, hist, C.byref(bn))
        return n, np.array(hist[:n + 1]), bn.value

    def nhydro_solve(self):
        hist = (C.c_double * (self.par.solver_maxiter + 1))()
        bn = C.c_double()
        n = lib().mgo_nhydro_solve(self.h, hist, C.byref(bn))
        return n, np.array(hist[:n + 1]), bn.value

    def check_nondivergence(self):
        lib().mgo_check_nondivergence(self.h)

    def fcycle(self):
        lib().mgo_fcycle(self.h)

    def vcycle(self, lev=1):
        lib().mgo_vcycle(self.h, lev)

    def relax(self, lev, nsweeps):
        lib().mgo_relax(self.h, lev, nsweeps)

    def residual(self, lev):
        return lib().mgo_residual(self.h, lev)

    def fine2coarse(self, lev):
        lib().mgo_fine2coarse(self.h, lev)

    def coarse2fine(self, lev):
        lib().mgo_coarse2fine(self.h, lev)

    def fill_halo(self, lev, name):
        lib().mgo_fill_halo(self.h, lev, FIELD[name])


# ---- synthetic inputs of the reference's drivers -------------------------
def seamount_geometry(nx, ny, npx, npy, rank, Lx=1e4, Ly=1e4, Htot=4e3):
    """mg_setup_tests.f90:108-158 setup_seamount for one rank: dx,dy,zeta,h as [i][j] arrays of (0:nx+1,0:ny+1)."""
    nxg, nyg = npx * nx, npy * ny
    pj, pi = rank // npx, rank % npx
    dxv, dyv = Lx / float(nxg), Ly / float(nyg)
    i = np.arange(0, nx + 2, dtype=np.float64)[:, None]
    j = np.arange(0, ny + 2, dtype=np.float64)[None, :]
    x = (i + pi * nx - 0.5) * dxv
    y = (j + pj * ny - 0.5) * dyv
    x0, y0 = Lx * 0.5, Ly * 0.5
    h = Htot * (1.0 - 0.5 * np.exp(-(x - x0) ** 2.0 / (Lx / 5.0) ** 2.0 - (y - y0) ** 2.0 / (Ly / 5.0) ** 2.0))
    dx = np.full((nx + 2, ny + 2), dxv)
    dy = np.full((nx + 2, ny + 2), dyv)
    zeta = np.zeros((nx + 2, ny + 2))
    return dx, dy, zeta, h


def rndtopo_geometry(nx, ny, npx, npy, rank, Lx=1e4, Ly=1e4, Htot=4e3, seed=12345):
    """Decomposition-independent random topography: h = Htot*0.2*U ("between 0% and 20% of Htot", mg_setup_tests.f90:199) drawn once per
    GLOBAL (i,j) from a seeded generator owned by this build, mirrored into the physical halo."""
    nxg, nyg = npx * nx, npy * ny
    pj, pi = rank // npx, rank % npx
    rng = np.random.Generator(np.random.PCG64(seed))
    hg = np.pad(Htot * 0.2 * rng.random((nxg, nyg)), 1, mode="edge")
    h = hg[pi * nx:pi * nx + nx + 2, pj * ny:pj * ny + ny + 2].copy()
    dx = np.full((nx + 2, ny + 2), Lx / float(nxg))
    dy = np.full((nx + 2, ny + 2), Ly / float(nyg))
    return dx, dy, np.zeros((nx + 2, ny + 2)), h


def make_seamount(nx, ny, nz, npx=1, npy=1, **kw):
    """mg_testseamount.f90:69-123: init, geometry, matrices, u=v=0, w=-1 (0 at the bottom)."""
    o = Oracle(nx, ny, nz, npx, npy, **kw)
    for r in range(o.nranks):
        dx, dy, zeta, h = seamount_geometry(nx, ny, npx, npy, r)
        o.field("dx", 1, r)[...] = dx
        o.field("dy", 1, r)[...] = dy
        o.field("zeta", 1, r)[...] = zeta
        o.field("h", 1, r)[...] = h
    o.matrices(4e3, 0.0, 0.0)
    for r in range(o.nranks):
        o.field("u", 1, r)[...] = 0.0
        o.field("v", 1, r)[...] = 0.0
        w = o.field("w", 1, r)
        w[0] = 0.0
        w[1:] = -1.0
    return o
