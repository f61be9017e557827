"""CPU tests of the product's host logic: library loads and exports every symbol of include/mgx.h, the level /
neighbour / gather tables equal the oracle's (mg_grids.f90:468-738), the namelist parser, loud failure without a GPU."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    import mgroms_amd
    return mgroms_amd


def test_every_declared_symbol_is_exported(built):
    from mgroms_amd._lib import lib, SYMBOLS
    hdr = open(os.path.join(ROOT, "include", "mgx.h")).read()
    declared = set(re.findall(r"\b(mgx_[a-z_0-9]+)\s*\(", hdr)) - {"mgx_set_comm"} | {"mgx_set_comm"}
    declared = {d for d in declared if not d.endswith("_fn")}
    assert declared == set(SYMBOLS), declared ^ set(SYMBOLS)
    L = lib()
    for s in declared:
        assert hasattr(L, s), s


def test_documented_options_exist(built):
    """Every run-time option include/mgx.h documents ("name" (default ...)) is known to the library: mgx_get_option succeeds (no GPU
    needed -- options are plain state), with the documented default for the round-4 red-black switches."""
    import ctypes
    from mgroms_amd._lib import lib
    hdr = open(os.path.join(ROOT, "include", "mgx.h")).read()
    names = set(re.findall(r'"([a-z_0-9]+)" \(default', hdr))
    assert {"rb_seq", "rb_exact", "rbseq_fuse", "rbseq_window", "rbseq_rowcut", "rbseq_fuse_min", "rbseq_d0_in_pass", "fuse_closing", "restrict_chain", "overlap", "async"} <= names, names
    L = lib()
    v = ctypes.c_int(-12345)
    for n in sorted(names - {"rbseq_timeout_ms", "ksp_timeout_ms", "p2p_timeout_ms"}):   # (device constants: write-only)
        assert L.mgx_get_option(n.encode(), ctypes.byref(v)) == 0, n
    for n, d in (("rb_seq", 1), ("rb_exact", 0), ("rbseq_fuse", 1), ("rbseq_window", 1), ("rbseq_fuse_min", 4 << 20), ("rbseq_d0_in_pass", 1)):
        assert L.mgx_get_option(n.encode(), ctypes.byref(v)) == 0 and v.value == d, (n, v.value)


@pytest.mark.parametrize("cfg", [(64, 64, 16, 1, 1, 8), (32, 32, 16, 2, 2, 8), (512, 512, 64, 1, 1, 8), (64, 32, 32, 4, 2, 8),
                                 (512, 512, 64, 2, 2, 8), (512, 1024, 128, 4, 2, 16), (128, 128, 64, 4, 2, 64), (16, 16, 8, 2, 1, 8),
                                 (16, 16, 8, 1, 2, 8), (32, 64, 8, 4, 4, 16)])
def test_level_tables_match_oracle(built, cfg):
    from mgroms_amd import nhydro
    from oracle.mgoracle import Oracle
    nx, ny, nz, npx, npy, nsmall = cfg
    small = nx * ny * nz * npx * npy <= 2 ** 22  # the oracle allocates every rank: only build it for small worlds
    if not small:
        t = nhydro.level_table(nx, ny, nz, npx, npy, 0, nsmall)
        assert t[0]["nx"] == nx and all(d["nx"] % 2 == 0 for d in t)
        return
    o = Oracle(nx, ny, nz, npx, npy, nsmall=nsmall)
    for rank in range(npx * npy):
        t = nhydro.level_table(nx, ny, nz, npx, npy, rank, nsmall)
        assert len(t) == o.nlevs
        for lev, d in enumerate(t, start=1):
            li = o.level_info(lev, rank)
            for k in ("nx", "ny", "nz", "npx", "npy", "incx", "incy", "gather", "ngx", "ngy", "neighb"):
                assert d[k] == li[k], (rank, lev, k)
            if d["gather"]:
                assert (d["key"], d["color"]) == (li["key"], li["color"])


def test_baseline_config_levels(built):
    from mgroms_amd import nhydro
    # 512x512x64 -> 6 levels down to 16x16x2; 2048x2048x128 on 4x2 -> 7 levels, no gather at nsmall=8
    t = nhydro.level_table(512, 512, 64)
    assert len(t) == 6 and (t[-1]["nx"], t[-1]["nz"]) == (16, 2)
    t = nhydro.level_table(512, 1024, 128, 4, 2, 0)
    assert len(t) == 7 and not any(d["gather"] for d in t)
    t = nhydro.level_table(512, 1024, 128, 4, 2, 0, nsmall=16)
    assert any(d["gather"] for d in t)


def test_namelist_parser(built, tmp_path):
    from mgroms_amd import nhydro
    from mgroms_amd._lib import MgxError
    f = tmp_path / "nh_namelist"
    f.write_text("!- comment\n&nhparam\n  solver_prec = 1.d-12, !- x\n  solver_maxiter = 7,\n  nsmall=16\n  relax_method = 'FC',\n"
                 "  cmatrix='simple', interp_type = 'nearest', netcdf_output = .true., bmask = .false.,\n/\n")
    p = nhydro.read_nhnamelist(str(f))
    assert (p.solver_prec, p.solver_maxiter, p.nsmall, p.relax_method, p.cmatrix, p.interp_type, p.netcdf_output, p.bmask) == \
        (1e-12, 7, 16, b"FC", b"simple", b"nearest", 1, 0)
    d = nhydro.read_nhnamelist(str(tmp_path / "missing"))  # absent file: defaults (mg_namelist.f90:75-86)
    assert (d.solver_prec, d.solver_maxiter, d.ns_coarsest, d.ns_pre, d.ns_post, d.relax_method) == (1e-6, 50, 40, 3, 2, b"RB")
    g = tmp_path / "stale"
    g.write_text("&nhparam\n nhalo = 1,\n/\n")  # the stale example in examples/namelist would abort a Fortran read too
    with pytest.raises(MgxError):
        nhydro.read_nhnamelist(str(g))
    h = tmp_path / "lin"
    h.write_text("&nhparam\n interp_type='linear', restrict_type='linear'\n/\n")
    with pytest.raises(MgxError):
        nhydro.read_nhnamelist(str(h))


def test_namelist_parser_against_reference_compiled_read(built, tmp_path):
    """mgx_read_namelist against read_nhnamelist of the reference itself (mg_namelist.f90 compiled unmodified with flang,
    oracle/Makefile target `ref`; fixtures tests/golden/ref_namelist.json made by oracle/make_ref_golden.py): defaults, every
    member set, Fortran `d` exponents, .true. / T, mixed case, comments, a one-line group, and the linear+linear rejection of
    :95-98 (which the reference shows at -O0 only: the branch reads `rank` uninitialised and flang -O2 deletes it)."""
    import json
    from mgroms_amd import nhydro
    from mgroms_amd._lib import MgxError
    with open(os.path.join(ROOT, "tests", "golden", "ref_namelist.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 8
    for name, c in cases.items():
        fn = tmp_path / name
        fn.write_text(c["text"])
        if not c["accepted"]:
            with pytest.raises(MgxError):
                nhydro.read_nhnamelist(str(fn))
            continue
        p = nhydro.read_nhnamelist(str(fn))
        m = c["members"]
        assert p.solver_prec == float(m["solver_prec"].replace("E-0", "E-").replace("E+0", "E+")), name
        for k in ("solver_maxiter", "nsmall", "ns_coarsest", "ns_pre", "ns_post"):
            assert getattr(p, k) == int(m[k]), (name, k)
        for k in ("cmatrix", "relax_method", "interp_type", "restrict_type"):
            assert getattr(p, k).decode() == m[k], (name, k)
        for k in ("aggressive", "netcdf_output", "bmask"):
            assert getattr(p, k) == (1 if m[k] == "T" else 0), (name, k)


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mgroms_amd import nhydro
    from mgroms_amd._lib import MgxError
    with pytest.raises(MgxError, match="no HIP device"):
        nhydro.nhydro_init(16, 16, 8)
    with pytest.raises(MgxError):
        nhydro.relax(1, 1)


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "mgroms_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(root, fn)).read()
                assert "oracle" not in txt.replace("the oracle", "").replace("CPU oracle", ""), fn


def test_bench_live_traffic_falls_back_without_a_gpu():
    """bench.py measures roofline.traffic with two rocprofv3 --pmc child passes of the same invocation; where that cannot be done (here: no GPU, so the
    profiled child fails) the helper returns None -- quickly -- and the line falls back to the committed capture, labelled as such."""
    import importlib
    import time
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    t0 = time.time()
    assert bench.live_traffic(512, 512, 64, "FC", "k_relax_nz<64, true, false, 3, true, true>") is None
    assert time.time() - t0 < 240
    os.environ["ROCPROF_TEST_MARK"] = "1"   # a bench that is itself being profiled must not start a profiler of its own
    try:
        t0 = time.time()
        assert bench.live_traffic(512, 512, 64, "FC", "k_relax_nz<64, true, false, 3, true, true>") is None
        assert time.time() - t0 < 1.0
    finally:
        del os.environ["ROCPROF_TEST_MARK"]


def test_windowed_walk_bounds_host_logic(built):
    """The two host-side decisions of the windowed red-black walk (mgx_rbseq.hip), callable without a GPU: the planes of warm-up from the contraction
    bound rho (rho^m <= 2^-64, at least 2, none beyond 48 or for a bound that is not a number below one) and the rows the correction reaches from the
    per-row decay figures (the last one above 2^-64; every row when a figure is not finite)."""
    import ctypes
    import math
    from mgroms_amd._lib import lib
    L = lib()
    planes = L.mgxk_rbseq_window_planes
    planes.restype = ctypes.c_int; planes.argtypes = [ctypes.c_double]
    for rho in (0.011, 0.025, 0.0385, 0.1, 0.2, 0.39):
        m = planes(rho)
        assert m >= 2 and rho ** m <= 2.0 ** -64 and (m == 2 or rho ** (m - 1) > 2.0 ** -64), (rho, m)
    assert planes(0.0385) == 14 and planes(0.0) == 1 and planes(1e-30) == 2
    assert planes(0.41) == 0 and planes(0.9) == 0 and planes(1.0) == 0 and planes(1.5) == 0 and planes(-0.1) == 0 and planes(float("nan")) == 0
    rows = L.mgxk_rbseq_window_rows
    rows.restype = ctypes.c_int; rows.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int]

    def nrows(v):
        a = (ctypes.c_double * len(v))(*v)
        return rows(a, len(v))
    dec = [0.02 ** k for k in range(64)]
    k = nrows(dec)
    assert k == 1 + max(i for i, d in enumerate(dec) if d > 2.0 ** -64) and k < 20
    assert nrows([1.0, 0.5, 0.25, 0.125]) == 4                      # nothing decays far enough: every row
    assert nrows([1.0] + [0.0] * 31) == 1
    assert nrows([1.0, 1e-30, float("inf"), 0.0]) == 4 and nrows([1.0, float("nan"), 0.0, 0.0]) == 4   # a figure that is not finite: no cut
    assert nrows([1.0, 1e-25, 1e-10, 1e-30]) == 3                  # (not monotone: the LAST row above the threshold counts)
