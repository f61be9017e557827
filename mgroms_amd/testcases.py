"""Synthetic inputs of the reference's drivers, restated for the harnesses (bench.py, scripts/, smoke()):
the analytic seamount of src/mg_setup_tests.f90:108-158 and the model state of src/mg_testseamount.f90:119-123.
Pure numpy; arrays use the package's convention (numpy C order = reversed Fortran index)."""
import numpy as np


def seamount_geometry(nx, ny, npx=1, npy=1, rank=0, Lx=1e4, Ly=1e4, Htot=4e3):
    """dx, dy, zeta, h of one rank as (nx+2, ny+2) arrays = Fortran (0:ny+1, 0:nx+1).
    h = Htot*(1 - 0.5*exp(-(x-x0)^2/(Lx/5)^2 - (y-y0)^2/(Ly/5)^2)), x = (i + pi*nx - 1/2)*dx (mg_setup_tests.f90:132-146)."""
    nxg, nyg = npx * nx, npy * ny
    pj, pi = rank // npx, rank % npx
    dxv, dyv = Lx / float(nxg), Ly / float(nyg)
    i = np.arange(0, nx + 2, dtype=np.float64)[:, None]
    j = np.arange(0, ny + 2, dtype=np.float64)[None, :]
    x = (i + pi * nx - 0.5) * dxv
    y = (j + pj * ny - 0.5) * dyv
    x0, y0 = Lx * 0.5, Ly * 0.5
    h = Htot * (1.0 - 0.5 * np.exp(-(x - x0) ** 2.0 / (Lx / 5.0) ** 2.0 - (y - y0) ** 2.0 / (Ly / 5.0) ** 2.0))
    return np.full((nx + 2, ny + 2), dxv), np.full((nx + 2, ny + 2), dyv), np.zeros((nx + 2, ny + 2)), h


def rndtopo_geometry(nx, ny, npx=1, npy=1, rank=0, Lx=1e4, Ly=1e4, Htot=4e3, seed=12345):
    """Decomposition-independent random topography (BASELINE config 4): h = Htot*0.2*U ("between 0% and 20% of Htot", mg_setup_tests.f90:199), one draw per GLOBAL
    (i,j) from a seeded generator, mirrored into the physical halo, then cut to this rank's block."""
    nxg, nyg = npx * nx, npy * ny
    pj, pi = rank // npx, rank % npx
    rng = np.random.Generator(np.random.PCG64(seed))
    hg = np.pad(Htot * 0.2 * rng.random((nxg, nyg)), 1, mode="edge")
    h = hg[pi * nx:pi * nx + nx + 2, pj * ny:pj * ny + ny + 2].copy()
    return (np.full((nx + 2, ny + 2), Lx / float(nxg)), np.full((nx + 2, ny + 2), Ly / float(nyg)),
            np.zeros((nx + 2, ny + 2)), h)


def island_mask(nx, ny, npx=1, npy=1, rank=0):
    """rmask of one rank for bmask=.true. runs, (nx+2, ny+2) = Fortran (0:ny+1, 0:nx+1): a closed basin (0 in the
    physical halo) with one round island, defined on GLOBAL indices so it does not depend on the decomposition."""
    nxg, nyg = npx * nx, npy * ny
    pj, pi = rank // npx, rank % npx
    ig = np.arange(0, nxg + 2, dtype=np.float64)[:, None]
    jg = np.arange(0, nyg + 2, dtype=np.float64)[None, :]
    m = np.ones((nxg + 2, nyg + 2))
    m[(ig - 0.3 * nxg) ** 2 + (jg - 0.6 * nyg) ** 2 <= (0.12 * min(nxg, nyg)) ** 2] = 0.0
    m[0, :] = m[-1, :] = 0.0
    m[:, 0] = m[:, -1] = 0.0
    return m[pi * nx:pi * nx + nx + 2, pj * ny:pj * ny + ny + 2].copy()


def resting_column_state(nx, ny, nz):
    """u = v = 0, w = -1 except 0 at the bottom (mg_testseamount.f90:119-123), in the model's (i,j,k) layout:
    numpy shapes (nz, ny+2, nx+1), (nz, ny+1, nx+2), (nz+1, ny+2, nx+2)."""
    u = np.zeros((nz, ny + 2, nx + 1))
    v = np.zeros((nz, ny + 1, nx + 2))
    w = -np.ones((nz + 1, ny + 2, nx + 2))
    w[0] = 0.0
    return u, v, w
