"""Generates tests/golden/ref_zrzw.npz, ref_namelist.json and ref_tictoc.txt from oracle/_ref/ref_driver, i.e. from the three
reference modules that compile UNMODIFIED here (mg_zr_zw.f90, mg_namelist.f90, mg_tictoc.f90; `make -C oracle ref`).

TEST INFRASTRUCTURE.  Run in the build container only (it needs /root/reference and flang); the fixtures it writes are data
(inputs and the reference's outputs) and travel with the repository.  Usage: python oracle/make_ref_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DRV = os.path.join(HERE, "_ref", "ref_driver")
GOLD = os.path.join(ROOT, "tests", "golden")


def geometry(nx, ny, kind):
    """h, zeta as [i][j] arrays of (0:nx+1, 0:ny+1) -- deterministic, no RNG state shared with anything else"""
    i = np.arange(nx + 2, dtype=np.float64)[:, None]
    j = np.arange(ny + 2, dtype=np.float64)[None, :]
    x, y = (i - 0.5) / nx, (j - 0.5) / ny
    h = 4e3 * (1.0 - 0.5 * np.exp(-(x - 0.5) ** 2 / 0.04 - (y - 0.5) ** 2 / 0.04))  # the seamount of mg_setup_tests.f90:145, unit square
    if kind == "rough":
        h = h * (0.6 + 0.4 * np.cos(1.7 * i + 0.3) * np.sin(0.9 * j + 1.1)) + 37.0
    zeta = np.zeros_like(h) if kind == "flat" else 0.4 * np.cos(0.25 * i) * np.sin(0.15 * j) + 0.05
    # physical-boundary halo = mirror of the first interior cell (fill_halo_2D, mg_mpi_exchange.f90:509-537), as define_matrices leaves
    # h and zeta before it calls setup_zr_zw (mg_define_matrix.f90:99-127): the whole 0:n+1 ring is then comparable
    mirror = lambda a: np.ascontiguousarray(np.pad(a[1:-1, 1:-1], 1, mode="edge"))
    return mirror(h), mirror(zeta)


ZRZW_CASES = [
    # name, nx, ny, nz, hlim, theta_b, theta_s, kind
    ("seamount_16x16x8_theta0", 16, 16, 8, 4e3, 0.0, 0.0, "flat"),          # BASELINE's set-up (mg_testseamount.f90:100-104)
    ("ragged_12x20x16_stretched", 12, 20, 16, 250.0, 0.4, 6.0, "rough"),      # cosh + exp branches, zeta /= 0
    ("col_8x8x64_stretched", 8, 8, 64, 250.0, 0.4, 6.0, "zeta"),              # nz = 64: the level-1 kernel's tables
    ("col_16x8x128_theta0_zeta", 16, 8, 128, 4e3, 0.0, 0.0, "zeta"),          # nz = 128 (config 5), moving free surface
    ("only_theta_s_8x8x4", 8, 8, 4, 100.0, 0.0, 3.0, "zeta"),                 # theta_s > 0, theta_b = 0
    ("only_theta_b_8x8x4", 8, 8, 4, 100.0, 0.7, 0.0, "rough"),                # theta_s = 0, theta_b > 0
]

NAMELISTS = {
    "defaults_empty_group": "&nhparam\n/\n",
    "reference_shipped": None,  # filled from the text below: the member values of the reference's src/nh_namelist, retyped
    "every_member": "&nhparam\n solver_prec = 2.5d-9,\n solver_maxiter = 17,\n nsmall = 16,\n ns_coarsest = 11,\n ns_pre = 4,\n ns_post = 1,\n"
                    " cmatrix = 'simple',\n relax_method = 'Four-Color',\n interp_type = 'nearest',\n restrict_type = 'linear',\n"
                    " netcdf_output = .true.,\n aggressive = .false.,\n bmask = .true.,\n/\n",
    "mixed_case_and_T": "&NHPARAM\n Solver_Prec = 1.E-8\n RELAX_METHOD = \"FC\"\n BMASK = T\n NetCDF_Output = F\n ns_pre=5 , ns_post=0\n/\n",
    "comments_and_d_exponent": "! a comment line\n&nhparam\n solver_prec = 1.D-12, ! trailing comment\n nsmall = 32 ! another\n relax_method = 'GS'\n/\n",
    "one_line": "&nhparam solver_maxiter=3, cmatrix='simple', relax_method='RB' /\n",
    "linear_linear_rejected": "&nhparam\n interp_type = 'linear',\n restrict_type = 'linear',\n/\n",
    "nearest_linear_allowed": "&nhparam\n interp_type = 'nearest',\n restrict_type = 'linear',\n/\n",
}
NAMELISTS["reference_shipped"] = ("&nhparam\n solver_prec = 1.d-12,\n solver_maxiter = 50,\n nsmall = 8,\n ns_coarsest = 40,\n ns_pre = 3,\n ns_post = 2,\n"
                                  " cmatrix = 'real',\n relax_method = 'RB',\n aggressive = .false.,\n interp_type = 'linear',\n restrict_type = 'avg',\n"
                                  " netcdf_output = .true.,\n bmask = .false.,\n/\n")


def run_zrzw(nx, ny, nz, hlim, tb, ts, h, zeta, tmp):
    fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
    with open(fin, "wb") as f:
        np.array([nx, ny, nz], dtype=np.int32).tofile(f)
        np.array([hlim, tb, ts], dtype=np.float64).tofile(f)
        h.tofile(f)       # [i][j] C order == Fortran (0:ny+1, 0:nx+1)
        zeta.tofile(f)
    subprocess.check_call([DRV, "zrzw", fin, fout])
    raw = np.fromfile(fout, dtype=np.float64)
    nzr = (nx + 4) * (ny + 4) * nz
    zr = raw[:nzr].reshape(nx + 4, ny + 4, nz)
    zw = raw[nzr:].reshape(nx + 4, ny + 4, nz + 1)
    return zr, zw


def run_namelist(text, tmp):
    fn = os.path.join(tmp, "nml")
    with open(fn, "w") as f:
        f.write(text)
    out = subprocess.run([DRV, "namelist", fn], capture_output=True, text=True, cwd=tmp)
    members = {}
    for line in out.stdout.splitlines():
        if " = " in line:
            k, v = line.split(" = ", 1)
            members[k.strip()] = v.strip()
    accepted = "end_of_members" in out.stdout
    return {"text": text, "accepted": accepted, "members": members if accepted else None}


def main():
    if not os.path.exists(DRV):
        subprocess.check_call(["make", "-C", HERE, "ref"])
    os.makedirs(GOLD, exist_ok=True)
    arrays = {}
    with tempfile.TemporaryDirectory(prefix="mgx_ref_") as tmp:
        for name, nx, ny, nz, hlim, tb, ts, kind in ZRZW_CASES:
            h, zeta = geometry(nx, ny, kind)
            zr, zw = run_zrzw(nx, ny, nz, hlim, tb, ts, h, zeta, tmp)
            arrays[name + "/par"] = np.array([nx, ny, nz, hlim, tb, ts])
            arrays[name + "/h"] = h
            arrays[name + "/zeta"] = zeta
            arrays[name + "/zr"] = zr
            arrays[name + "/zw"] = zw
        np.savez_compressed(os.path.join(GOLD, "ref_zrzw.npz"), **arrays)
        nml = {k: run_namelist(v, tmp) for k, v in NAMELISTS.items()}
        with open(os.path.join(GOLD, "ref_namelist.json"), "w") as f:
            json.dump({"_source": "read_nhnamelist of /root/reference/src/mg_namelist.f90 compiled unmodified (oracle/Makefile target ref), "
                                  "driven by oracle/ref_driver.f90 via oracle/make_ref_golden.py", "cases": nml}, f, indent=1)
        tt = os.path.join(tmp, "tictoc.txt")
        subprocess.check_call([DRV, "tictoc", tt], cwd=tmp)
        with open(tt) as f, open(os.path.join(GOLD, "ref_tictoc.txt"), "w") as g:
            g.write(f.read())
    print("wrote", sorted(os.listdir(GOLD)))


if __name__ == "__main__":
    sys.exit(main())
