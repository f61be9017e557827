"""GPU parity tests: the HIP path, driven through the C ABI (mgroms_amd -> libmgx.so), against the CPU oracle on
the same inputs, and against the reference's own known answers (tests/golden).

Tolerances.  Four-colour ('FC') smoothing, the residual, the transfers and the coefficient set-up keep the
reference's operation order and are compiled without FMA contraction, so their fields are compared EXACTLY
(bit for bit) with the oracle.  Norms are reduced in a different order on the GPU: 1e-13 relative.  Residual
histories against the ORACLE (same fields, other reduction order): 1e-12 relative; against the reference's RECORDED series
(tests/golden: another build, its libm): |d| <= 1e-13 + 1e-10*ref in units of ||b|| (north_star: 1e-10 relative).  Red-black ('RB') is order
dependent in the reference itself.  The default ("rb_seq") reproduces the reference's sequential order up to
the association of one sum per column: fields within 1e-12 of the oracle's sequential loop per relax call, histories within
north_star's 1e-10; "rb_exact" is the same order bit for bit (one launch per plane); the plain parallel sweep (rb_seq = 0) is
checked at the tolerance at which the reference agrees with itself across decompositions (5e-5 relative)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    import mgroms_amd as m
    m.nhydro.set_verbose(0)
    yield m
    m.nhydro_clean()


def _setup(mg, nx, ny, nz, geom="seamount", **par):
    from oracle.mgoracle import Oracle, seamount_geometry, rndtopo_geometry
    kw = dict(relax_method="FC", solver_prec=1e-10)
    kw.update(par)
    p = mg.nhydro.default_params(**kw)
    mg.nhydro_init(nx, ny, nz, 1, 1, 0, p)
    g = seamount_geometry if geom == "seamount" else rndtopo_geometry
    dx, dy, zeta, h = g(nx, ny, 1, 1, 0)
    mg.nhydro_matrices(dx, dy, zeta, h, None, 4e3, 0.0, 0.0)
    okw = {k: v for k, v in kw.items()}
    o = Oracle(nx, ny, nz, 1, 1, **okw)
    for name, a in (("dx", dx), ("dy", dy), ("zeta", zeta), ("h", h)):
        o.field(name)[...] = a
    o.matrices(4e3, 0.0, 0.0)
    return o


def _uvw(nx, ny, nz, seed=None):
    if seed is None:
        u = np.zeros((nz, ny + 2, nx + 1)); v = np.zeros((nz, ny + 1, nx + 2)); w = -np.ones((nz + 1, ny + 2, nx + 2)); w[0] = 0
    else:
        r = np.random.default_rng(seed)
        u = r.uniform(-1, 1, (nz, ny + 2, nx + 1)); v = r.uniform(-1, 1, (nz, ny + 1, nx + 2)); w = r.uniform(-1, 1, (nz + 1, ny + 2, nx + 2))
    return u, v, w


@pytest.mark.parametrize("dims,geom", [((16, 16, 8), "seamount"), ((64, 32, 16), "seamount"), ((32, 32, 8), "rndtopo")])
def test_define_matrices_bitwise(mg, dims, geom):
    nx, ny, nz = dims
    o = _setup(mg, nx, ny, nz, geom)
    assert mg.nlevs() == o.nlevs
    for lev in range(1, o.nlevs + 1):
        g = mg.grid(lev)
        li = o.level_info(lev)
        assert (g.nx, g.ny, g.nz) == (li["nx"], li["ny"], li["nz"])
        for name in ("dx", "dy", "h", "zr", "zw", "cw", "cA"):
            a, b = g.get(name), o.field(name, lev)
            assert np.array_equal(a, b), (lev, name, np.abs(a - b).max())


def test_residual_and_norm(mg):
    nx, ny, nz = 32, 16, 8
    o = _setup(mg, nx, ny, nz)
    r = np.random.default_rng(1)
    for lev in (1, 2):
        g = mg.grid(lev)
        p = r.standard_normal(g._shape("p")); b = r.standard_normal(g._shape("b"))
        g.set("p", p); g.set("b", b)
        mg.fill_halo(lev, "p")
        o.field("p", lev)[...] = p; o.field("b", lev)[...] = b
        o.fill_halo(lev, "p")
        assert np.array_equal(g.get("p"), o.field("p", lev))  # halo rules (mirror + corners)
        res = mg.compute_residual(lev)
        reso = o.residual(lev)
        assert np.array_equal(g.get("r"), o.field("r", lev))
        assert abs(res - reso) <= 1e-13 * reso


def test_testgalerkin_through_the_abi(mg):
    # testgalerkin (mg_solvers.f90:203-288) through the C ABI against the same sequence on the oracle: a smooth coarse field,
    # <xc, Ac xc> and <I xc, Af I xc> (the reference prints norm_c, norm_f/4 and norm_c/norm_f*4); the two energies agree within 10 %
    o = _setup(mg, 32, 32, 16)
    i, j, k = np.meshgrid(np.arange(16), np.arange(16), np.arange(8), indexing="ij")
    xc = np.sin(np.pi * (i + 0.5) / 16) ** 2 * np.sin(np.pi * (j + 0.5) / 16) ** 2 * np.sin(np.pi * (k + 0.5) / 8)
    pc = np.zeros(mg.grid(2)._shape("p")); pc[1:-1, 1:-1, :] = xc
    mg.grid(2).set("p", pc)
    nc, nf = mg.nhydro.testgalerkin(2)
    o.field("p", 2)[...] = pc; o.fill_halo(2, "p"); o.field("b", 2)[...] = 0; o.residual(2)
    nco = (o.field("p", 2)[1:-1, 1:-1, :] * o.field("r", 2)[1:-1, 1:-1, :]).sum()
    o.field("p", 1)[...] = 0; o.coarse2fine(1); o.field("b", 1)[...] = 0; o.residual(1)
    nfo = (o.field("p", 1)[1:-1, 1:-1, :] * o.field("r", 1)[1:-1, 1:-1, :]).sum()
    assert abs(nc - nco) <= 1e-12 * abs(nco) and abs(nf - nfo) <= 1e-12 * abs(nfo)
    assert np.array_equal(mg.grid(1).p, o.field("p", 1))
    assert 0.85 < nf / nc < 1.1


def test_generic_fill_halo(mg):
    # fill_halo(lev, field) for the array kinds beyond p/b/r (mg_mpi_exchange.f90:10-16): 2-D mirror, nh=2 extrapolation (zr, zw)
    o = _setup(mg, 32, 16, 8)
    r = np.random.default_rng(12)
    for lev in (1, 2):
        g = mg.grid(lev)
        for name in ("dx", "h", "zr", "zw"):
            a = r.standard_normal(g._shape(name))
            g.set(name, a); o.field(name, lev)[...] = a
            mg.fill_halo(lev, name); o.fill_halo(lev, name)
            assert np.array_equal(g.get(name), o.field(name, lev)), (lev, name)
    from mgroms_amd._lib import MgxError
    with pytest.raises(MgxError):
        mg.fill_halo(1, "cw")


@pytest.mark.parametrize("cmatrix", ["real", "simple"])
def test_relax_fc_bitwise(mg, cmatrix):
    nx, ny, nz = 32, 32, 16
    o = _setup(mg, nx, ny, nz, cmatrix=cmatrix)
    r = np.random.default_rng(2)
    for lev in (1, 2, o.nlevs):
        g = mg.grid(lev)
        p = r.standard_normal(g._shape("p")); b = r.standard_normal(g._shape("b"))
        g.set("p", p); g.set("b", b); mg.fill_halo(lev, "p")
        o.field("p", lev)[...] = p; o.field("b", lev)[...] = b; o.fill_halo(lev, "p")
        mg.relax(lev, 2)
        o.relax(lev, 2)
        assert np.array_equal(g.get("p"), o.field("p", lev)), lev


def test_relax_rb_simple_bitwise(mg):
    # without the k=1 horizontal diagonals (cmatrix='simple') red-black is order independent: exact parity
    nx, ny, nz = 32, 32, 16
    o = _setup(mg, nx, ny, nz, relax_method="RB", cmatrix="simple")
    r = np.random.default_rng(4)
    for lev in (1, 2, o.nlevs):
        g = mg.grid(lev)
        p = r.standard_normal(g._shape("p")); b = r.standard_normal(g._shape("b"))
        g.set("p", p); g.set("b", b); mg.fill_halo(lev, "p")
        o.field("p", lev)[...] = p; o.field("b", lev)[...] = b; o.fill_halo(lev, "p")
        mg.relax(lev, 2)
        o.relax(lev, 2)
        assert np.array_equal(g.get("p"), o.field("p", lev)), lev


def test_relax_rb_chained_snapshot_is_exact(mg):
    # cmatrix='real' red-black reads same-colour k=1 diagonals from a snapshot taken before the pass.  On a closed level the
    # colour passes chain that snapshot themselves (one launch per relax call): must equal a snapshot launch per pass, bit for bit
    nx, ny, nz = 64, 32, 16
    _setup(mg, nx, ny, nz, relax_method="RB")
    mg.nhydro.set_option("rb_seq", 0)   # the plain parallel sweep is what chains its snapshots
    r = np.random.default_rng(21)
    out = []
    for chain in (1, 0):
        mg.nhydro.set_option("rb_chain", chain)
        for lev in (1, 2):
            g = mg.grid(lev)
            rr = np.random.default_rng(21 + lev)
            g.set("p", rr.standard_normal(g._shape("p"))); g.set("b", rr.standard_normal(g._shape("b")))
            mg.fill_halo(lev, "p")
            mg.relax(lev, 3)
            out.append(g.get("p"))
    mg.nhydro.set_option("rb_chain", 1)
    mg.nhydro.set_option("rb_seq", 1)
    assert np.array_equal(out[0], out[2]) and np.array_equal(out[1], out[3])
    assert np.abs(out[0]).max() > 0


def test_relax_rb_real_close(mg):
    # with cmatrix='real' a red column reads its four same-colour diagonal neighbours at k=1: the sequential sweep
    # sees some of them updated, the parallel sweep sees all of them old -> O(diagonal coupling) difference only
    nx, ny, nz = 32, 32, 16
    o = _setup(mg, nx, ny, nz, relax_method="RB")
    r = np.random.default_rng(6)
    g = mg.grid(1)
    p = r.standard_normal(g._shape("p")); b = r.standard_normal(g._shape("b"))
    g.set("p", p); g.set("b", b); mg.fill_halo(1, "p")
    o.field("p")[...] = p; o.field("b")[...] = b; o.fill_halo(1, "p")
    mg.nhydro.set_option("rb_seq", 0)
    try:
        mg.relax(1, 1)
    finally:
        mg.nhydro.set_option("rb_seq", 1)
    o.relax(1, 1)
    a, c = g.get("p"), o.field("p")
    assert np.abs(a - c).max() <= 0.1 * np.abs(c).max()  # random (rough) fields: the diagonal terms are not small
    assert np.abs(a - c).max() > 0
    # the same sweep in the sequential order (the default): the rough field's diagonal terms are reproduced, not neglected
    g.set("p", p); mg.fill_halo(1, "p")
    mg.relax(1, 1)
    assert np.abs(g.get("p") - c).max() <= 1e-12 * np.abs(c).max()


@pytest.mark.parametrize("dims,geom", [((32, 32, 16), "seamount"), ((64, 32, 16), "rndtopo"), ((16, 128, 8), "seamount"), ((128, 256, 32), "rndtopo"),
                                       ((256, 512, 8), "seamount"), ((48, 96, 16), "rndtopo"), ((8, 1024, 4), "seamount")])
def test_relax_rb_sequential_order_at_speed(mg, dims, geom):
    """relax_method='RB', cmatrix='real' -- the reference default -- in the reference's SEQUENTIAL order (mg_relax.f90:170-186) without
    a launch per plane (option "rb_seq", the default; mgx_rbseq.hip): parallel colour pass, one-wave walk over the planes for the k=1
    couplings, rank-one correction per column.  Against the oracle's sequential loop from a rough random state (where the
    same-colour diagonal terms are large), three sweeps on every level: 1e-12 of max|p| (a few ulp per sweep); every half-row
    width the walk is instantiated for (partial wave, 1, 2, 4, 8 columns per lane), ragged blocks, both geometries."""
    nx, ny, nz = dims
    o = _setup(mg, nx, ny, nz, geom, relax_method="RB")
    assert mg.nhydro.get_option("rb_seq") == 1
    rng = np.random.default_rng(31)
    for lev in range(1, o.nlevs + 1):
        g = mg.grid(lev)
        p = rng.standard_normal(g._shape("p")); b = rng.standard_normal(g._shape("b"))
        g.set("p", p); g.set("b", b); mg.fill_halo(lev, "p")
        o.field("p", lev)[...] = p; o.field("b", lev)[...] = b; o.fill_halo(lev, "p")
        mg.relax(lev, 3); o.relax(lev, 3)
        a, c = g.get("p"), o.field("p", lev)
        assert np.abs(a - c).max() <= 1e-12 * np.abs(c).max(), (lev, np.abs(a - c).max() / np.abs(c).max())


@pytest.mark.parametrize("dims", [(32, 512, 16), (64, 256, 32), (48, 128, 8), (16, 256, 8), (144, 256, 8)])
def test_rb_sequential_order_correction_inside_the_walk_launch(mg, dims):
    """Option "rbseq_fuse" (default 1, used from "rbseq_fuse_min" cells of a colour on -- set to 0 here so that small levels take it): the
    per-column correction of the sequential-order red-black runs inside the launch of the walk over the planes, chasing it (workers on the
    other XCDs, u handed over by forwarding waves: mgx_rbseq.hip, k_rbseq_scan FUSE).  The same bits as the correction in a launch of its
    own, three sweeps per level from a rough random state, six repetitions (the hand-off is a cross-XCD publish: a stale read would show
    as a different field in some repetition); one launch fewer per colour where an instance exists (half-rows of 256 columns; with
    half-rows of at most 64 columns and at most 128 planes the same option selects k_rbseq_walk_apply instead: every workgroup redoes the walk up to its planes
    and corrects them, no hand-off -- the third shape and the coarse levels of the others); and within 1e-12 of the oracle's sequential loop.
    The last two shapes have two workers per plane (half-rows of 128 columns, 8 rows), so a workgroup's four workers lie in two planes and,
    every fourth workgroup, on either side of a chunk boundary: such a workgroup has to wait for BOTH chunks' words (they are set by different
    forwarding waves in no particular order -- waiting for the last worker's chunk only was a race that showed once in a few runs)."""
    nx, ny, nz = dims
    o = _setup(mg, nx, ny, nz, "seamount", relax_method="RB")
    rng = np.random.default_rng(37)
    init = {}
    for lev in range(1, o.nlevs + 1):
        g = mg.grid(lev)
        init[lev] = (rng.standard_normal(g._shape("p")), rng.standard_normal(g._shape("b")))

    def run(fuse):
        mg.nhydro.set_option("rbseq_fuse", fuse)
        out, launches = {}, {}
        for lev in range(1, o.nlevs + 1):
            g = mg.grid(lev)
            g.set("p", init[lev][0]); g.set("b", init[lev][1]); mg.fill_halo(lev, "p")
            n0 = mg.nhydro.counters()["launches"]
            mg.relax(lev, 3)
            launches[lev] = mg.nhydro.counters()["launches"] - n0
            out[lev] = g.get("p")
        return out, launches

    try:
        mg.nhydro.set_option("rbseq_fuse_min", 0)
        mg.nhydro.set_option("rbseq_window", 0)   # (the windowed walk would serve these levels: this test is about the walk over the whole level)
        ref, lref = run(0)
        fused_levels = 0
        for rep in range(6 if nx * ny * nz > 100000 else 24):
            got, lgot = run(1)
            for lev in ref:
                assert np.array_equal(got[lev], ref[lev]), (rep, lev, np.abs(got[lev] - ref[lev]).max())
            fused_levels = sum(1 for lev in ref if lgot[lev] == lref[lev] - 6)   # 3 sweeps x 2 colours, one launch fewer each
        assert fused_levels >= 1, (lref, lgot)
    finally:
        mg.nhydro.set_option("rbseq_fuse", 1); mg.nhydro.set_option("rbseq_fuse_min", 4 << 20); mg.nhydro.set_option("rbseq_window", 1)
    lev = 1
    o.field("p", lev)[...] = init[lev][0]; o.field("b", lev)[...] = init[lev][1]; o.fill_halo(lev, "p")
    o.relax(lev, 3)
    c = o.field("p", lev)
    assert np.abs(ref[lev] - c).max() <= 1e-12 * np.abs(c).max()


def test_rb_sequential_order_fused_launch_bounded_waits(mg):
    """The waits inside the fused walk + correction launch are bounded: with the test hook "rbseq_test_stall" the walk keeps its progress
    to itself, the forwarding waves give up after "rbseq_timeout_ms", release every word (the launch drains) and raise the error word:
    the next synchronising call fails loudly and turns the fused launch off; the same call then runs with the correction in a launch
    of its own and gives the sequential-order result."""
    from mgroms_amd._lib import MgxError
    nx, ny, nz = 32, 512, 16     # half-rows of 256 columns: the level-1 walk with forwarding waves and workers
    o = _setup(mg, nx, ny, nz, "seamount", relax_method="RB")
    rng = np.random.default_rng(43)
    g = mg.grid(1)
    p0 = rng.standard_normal(g._shape("p")); b0 = rng.standard_normal(g._shape("b"))
    g.set("b", b0); g.set("p", p0); mg.fill_halo(1, "p")
    o.field("p")[...] = p0; o.field("b")[...] = b0; o.fill_halo(1, "p")
    o.relax(1, 2)
    mg.nhydro.set_option("rbseq_fuse_min", 0)
    mg.nhydro.set_option("rbseq_window", 0)
    mg.nhydro.set_option("rbseq_timeout_ms", 50)
    mg.nhydro.set_option("rbseq_test_stall", 1)
    try:
        with pytest.raises(MgxError, match="lost its hand-off"):
            mg.relax(1, 2)
            g.get("p")
        assert mg.nhydro.get_option("rbseq_fuse") == 0
        g.set("p", p0); mg.fill_halo(1, "p")
        mg.relax(1, 2)
        c = o.field("p")
        assert np.abs(g.get("p") - c).max() <= 1e-12 * np.abs(c).max()
    finally:
        mg.nhydro.set_option("rbseq_timeout_ms", 2000)
        mg.nhydro.set_option("rbseq_fuse", 1); mg.nhydro.set_option("rbseq_fuse_min", 4 << 20); mg.nhydro.set_option("rbseq_window", 1)


@pytest.mark.parametrize("dims,geom", [((32, 512, 16), "seamount"), ((64, 256, 32), "rndtopo"), ((48, 96, 16), "seamount"), ((16, 1024, 8), "seamount"),
                                       ((144, 256, 8), "rndtopo"), ((48, 160, 16), "rndtopo"), ((20, 160, 4), "seamount"),
                                       ((16, 128, 64), "seamount"), ((16, 64, 128), "seamount")])
def test_rb_sequential_order_windowed_walk(mg, dims, geom):
    """Option "rbseq_window" (default 1): walk and correction of a colour in one launch without a walk over the whole level -- every workgroup
    walks the m planes in front of its own over its chunk of columns +- 32, from zero (mgx_rbseq.hip: k_rbseq_window).  m comes from the
    level's contraction bound rho = max |ag5| + |ag8| (found when the coefficients are built): rho^m <= 2^-64.  Checked here: the bound
    against the same maximum formed from the oracle's coefficients, m against it, the colours really done that way, and three sweeps per level
    from a rough random state against the oracle's sequential loop (1e-12 of max|p|) AND against the walk over the whole level
    ("rbseq_window" = 0: 1e-14 -- a truncation of 2^-64 of the largest increment per colour).  Shapes: half-rows of 256 / 128 / 48 / 512 columns, ragged
    ones (80 = a full chunk + 16 columns, 40), a plane count that is not a multiple of 8 (plain block order), nz = 4 (one row per wave), tall columns (nz = 64 and 128: the correction is cut
    where g has decayed; the nz = 128 colour pass leaves no d0, k_rbseq_d0 runs)."""
    nx, ny, nz = dims
    o = _setup(mg, nx, ny, nz, geom, relax_method="RB")
    rng = np.random.default_rng(47)
    assert mg.nhydro.get_option("rbseq_window") == 1
    for lev in range(1, o.nlevs + 1):
        g = mg.grid(lev)
        rho, m = mg.nhydro.rbseq_window_info(lev)
        # the same bound from the oracle's coefficients: g1 = (T^-1 e1)(1) by tridiag's recurrences (mg_relax.f90:308-334), times cA(5|8,1,j,i)
        cA = o.field("cA", lev)[1:-1, 1:-1]
        d, dd = cA[..., 0], cA[..., 1]
        n = d.shape[-1]
        bet = 1.0 / d[..., 0]; x = np.zeros_like(d); gam = np.zeros_like(d); x[..., 0] = bet
        for k in range(1, n):
            gam[..., k] = dd[..., k] * bet
            bet = 1.0 / (d[..., k] - dd[..., k] * gam[..., k])
            x[..., k] = (0 - dd[..., k] * x[..., k - 1]) * bet
        for k in range(n - 2, -1, -1):
            x[..., k] -= gam[..., k + 1] * x[..., k + 1]
        rho_o = (np.abs(x[..., 0] * cA[..., 0, 4]) + np.abs(x[..., 0] * cA[..., 0, 7])).max()
        assert abs(rho - rho_o) <= 1e-12 * rho_o, (lev, rho, rho_o)
        assert m >= 2 and rho ** m <= 2.0 ** -64 and (m == 2 or rho ** (m - 1) > 2.0 ** -64), (lev, rho, m)
        p = rng.standard_normal(g._shape("p")); b = rng.standard_normal(g._shape("b"))
        o.field("p", lev)[...] = p; o.field("b", lev)[...] = b; o.fill_halo(lev, "p")
        o.relax(lev, 3)
        c = o.field("p", lev)
        res = {}
        for win in (0, 1):
            mg.nhydro.set_option("rbseq_window", win)
            try:
                g.set("p", p); g.set("b", b); mg.fill_halo(lev, "p")
                n0 = mg.nhydro.get_option("rbseq_window_colours")
                mg.relax(lev, 3)
                res[win] = (g.get("p"), mg.nhydro.get_option("rbseq_window_colours") - n0)
            finally:
                mg.nhydro.set_option("rbseq_window", 1)
        assert res[0][1] == 0
        # 3 sweeps x 2 colours; the one-workgroup levels (the coarsest ones) run the plane loop inside k_relax_wave instead
        assert res[1][1] == 6 or (lev >= 2 and res[1][1] == 0 and g.nx * g.ny * g.nz <= 8192), (lev, res[1][1], g.nx, g.ny, g.nz)
        assert np.abs(res[1][0] - c).max() <= 1e-12 * np.abs(c).max(), (lev, np.abs(res[1][0] - c).max() / np.abs(c).max())
        assert np.abs(res[1][0] - res[0][0]).max() <= 1e-14 * np.abs(c).max(), (lev, np.abs(res[1][0] - res[0][0]).max() / np.abs(c).max())
        # the correction stops at the last row it reaches to 2^-64 ("rbseq_rowcut"): g = T^-1 e1 decays away from the bottom row -- whether a level is cut depends on
        # dz / dx (test_rb_windowed_walk_stops_where_the_correction_has_decayed builds a case that is); against every row corrected: 1e-14
        rows = mg.nhydro.rbseq_window_rows(lev)
        gk = np.abs(x / x[..., :1]).reshape(-1, n).max(0)
        assert 1 <= rows <= g.nz, (lev, rows)
        assert rows == g.nz or gk[rows:].max() <= 2.0 ** -64, (lev, rows, gk)
        assert rows == 1 or gk[rows - 1] > 2.0 ** -64, (lev, rows, gk)
        if res[1][1] == 6 and rows < g.nz:
            mg.nhydro.set_option("rbseq_rowcut", 0)
            try:
                g.set("p", p); g.set("b", b); mg.fill_halo(lev, "p")
                mg.relax(lev, 3)
                full = g.get("p")
            finally:
                mg.nhydro.set_option("rbseq_rowcut", 1)
            assert np.abs(res[1][0] - full).max() <= 1e-14 * np.abs(c).max(), (lev, np.abs(res[1][0] - full).max() / np.abs(c).max())


@pytest.mark.parametrize("dims", [(16, 128, 64), (16, 64, 128)])
def test_rb_windowed_walk_stops_where_the_correction_has_decayed(mg, dims):
    """Option "rbseq_rowcut" (default 1) on columns that are tall against their width (the seamount's geometry with dx, dy scaled to 19 m and the mount flattened
    accordingly: the cell shape of 512x512x64 and of BASELINE config 5): g = T^-1 e1 decays by a few per cent per row, the bound max |g(k) / g(1)| falls below 2^-64 well inside the
    column, and the windowed walk's correction neither reads nor writes the rows above (mgx_rbseq_window_rows < nz, checked against the same figure
    from the oracle's coefficients).  Three sweeps from a rough random state on level 1 against the oracle's sequential loop (1e-12) and against every
    row corrected (1e-14); nz = 128: the colour pass (k_relax_tall) leaves no d0, k_rbseq_d0 runs in front of the window launch."""
    from oracle.mgoracle import Oracle, seamount_geometry
    nx, ny, nz = dims
    mg.nhydro_init(nx, ny, nz, 1, 1, 0, mg.nhydro.default_params(relax_method="RB"))
    dx, dy, zeta, h = seamount_geometry(nx, ny, 1, 1, 0)
    dx = dx * (19.5 / dx.max()); dy = dy * (19.5 / dy.max())
    h = 4e3 - 0.01 * (4e3 - h)     # ... and the mount flattened with it (the slopes of the real grid, not of a 2 km mount across 300 m)
    mg.nhydro_matrices(dx, dy, zeta, h, None, 4e3, 0.0, 0.0)
    o = Oracle(nx, ny, nz, 1, 1, relax_method="RB")
    for name, a in (("dx", dx), ("dy", dy), ("zeta", zeta), ("h", h)):
        o.field(name)[...] = a
    o.matrices(4e3, 0.0, 0.0)
    cA = o.field("cA", 1)[1:-1, 1:-1]
    d, dd = cA[..., 0], cA[..., 1]
    bet = 1.0 / d[..., 0]; x = np.zeros_like(d); gam = np.zeros_like(d); x[..., 0] = bet
    for k in range(1, nz):
        gam[..., k] = dd[..., k] * bet
        bet = 1.0 / (d[..., k] - dd[..., k] * gam[..., k])
        x[..., k] = (0 - dd[..., k] * x[..., k - 1]) * bet
    for k in range(nz - 2, -1, -1):
        x[..., k] -= gam[..., k + 1] * x[..., k + 1]
    gk = np.abs(x / x[..., :1]).reshape(-1, nz).max(0)
    rows = mg.nhydro.rbseq_window_rows(1)
    assert 1 < rows < nz // 2 and gk[rows:].max() <= 2.0 ** -64 < gk[rows - 1], (rows, gk[:rows + 2])
    rng = np.random.default_rng(61)
    g = mg.grid(1)
    p = rng.standard_normal(g._shape("p")); b = rng.standard_normal(g._shape("b"))
    o.field("p")[...] = p; o.field("b")[...] = b; o.fill_halo(1, "p")
    o.relax(1, 3)
    c = o.field("p")
    res = {}
    for cut in (1, 0):
        mg.nhydro.set_option("rbseq_rowcut", cut)
        try:
            g.set("p", p); g.set("b", b); mg.fill_halo(1, "p")
            n0 = mg.nhydro.get_option("rbseq_window_colours")
            mg.relax(1, 3)
            assert mg.nhydro.get_option("rbseq_window_colours") - n0 == 6
            res[cut] = g.get("p")
        finally:
            mg.nhydro.set_option("rbseq_rowcut", 1)
    assert np.abs(res[1] - c).max() <= 1e-12 * np.abs(c).max(), np.abs(res[1] - c).max() / np.abs(c).max()
    assert np.abs(res[1] - res[0]).max() <= 1e-14 * np.abs(c).max(), np.abs(res[1] - res[0]).max() / np.abs(c).max()


def test_rb_sequential_order_window_refused_when_the_walk_contracts_slowly(mg):
    """A matrix whose same-colour couplings are strong (the oracle's coefficients with cA(5,1,:,:), cA(8,1,:,:) scaled up through set_field('cA')
    until rho > 0.39): the bound asks for more than 48 planes of warm-up, the windowed walk is NOT used (planes = 0, no colour counted) and the
    walk over the whole level gives the oracle's sequential-order result with the same matrix."""
    nx, ny, nz = 32, 64, 16
    o = _setup(mg, nx, ny, nz, "seamount", relax_method="RB")
    rho0, m0 = mg.nhydro.rbseq_window_info(1)
    assert 0 < rho0 < 0.1 and m0 > 0
    f = 0.6 / rho0
    cA = o.field("cA", 1)
    cA[:, :, 0, 4] *= f; cA[:, :, 0, 7] *= f
    mg.grid(1).set("cA", cA)
    rho, m = mg.nhydro.rbseq_window_info(1)
    assert abs(rho - 0.6) < 1e-9 and m == 0, (rho, m)
    rng = np.random.default_rng(53)
    g = mg.grid(1)
    p = rng.standard_normal(g._shape("p")); b = rng.standard_normal(g._shape("b"))
    g.set("p", p); g.set("b", b); mg.fill_halo(1, "p")
    o.field("p")[...] = p; o.field("b")[...] = b; o.fill_halo(1, "p")
    n0 = mg.nhydro.get_option("rbseq_window_colours")
    mg.relax(1, 2); o.relax(1, 2)
    assert mg.nhydro.get_option("rbseq_window_colours") == n0
    c = o.field("p")
    assert np.abs(g.get("p") - c).max() <= 1e-12 * np.abs(c).max()


@pytest.mark.parametrize("method,dims,geom", [("RB", (64, 64, 16), "seamount"), ("RB", (128, 256, 32), "rndtopo"), ("FC", (128, 128, 16), "seamount"),
                                              ("FC", (64, 128, 32), "rndtopo"), ("RB", (32, 32, 4), "seamount")])
def test_coarsest_solve_as_one_matrix_vector_product(mg, method, dims, geom):
    """Option "coarsest_direct": inside a cycle the coarsest level is entered with p = 0 and left after ns_coarsest sweeps -- a fixed linear map of b,
    whose matrix the level's own relax kernel builds from the unit vectors (lazily, after the coefficients changed) and which then costs one
    matrix-vector product (mgx_relax_coarse.hip: k_coarse_direct).  Default 1: used for red-black in the sequential order (a tolerance-based iteration
    anyway), NOT for four colours (bit parity kept).  Checked: two cycles from a random right-hand side with the option off, at its default and forced
    (2) -- every level's p within 1e-12 of the sweeps' (the same map in another association), bit for bit where the default must not use it; the
    counter of direct solves; a rebuild after the matrix changed (set_field('cA') of the coarsest level: scaled coefficients, another map); and the
    oracle's two cycles within 1e-12."""
    nx, ny, nz = dims
    o = _setup(mg, nx, ny, nz, geom, relax_method=method)
    rng = np.random.default_rng(59)
    g1 = mg.grid(1)
    b = rng.standard_normal(g1._shape("b"))
    o.field("b")[...] = b; o.field("p")[...] = 0.0
    o.vcycle(1); o.vcycle(1)

    def run(opt):
        mg.nhydro.set_option("coarsest_direct", opt)
        try:
            g1.set("b", b); g1.set("p", np.zeros(g1._shape("p")))
            n0 = mg.nhydro.get_option("coarsest_direct_solves")
            mg.Vcycle(1); mg.Vcycle(1)
            return [mg.grid(l).get("p") for l in range(1, o.nlevs + 1)], mg.nhydro.get_option("coarsest_direct_solves") - n0
        finally:
            mg.nhydro.set_option("coarsest_direct", 1)

    off, n_off = run(0)
    dflt, n_dflt = run(1)
    forced, n_forced = run(2)
    assert n_off == 0 and n_forced == 2 and n_dflt == (2 if method == "RB" else 0), (n_off, n_dflt, n_forced)
    tol_o = 1e-12 if method == "FC" else 1e-10    # (the sequential-order red-black is itself a few ulp per sweep from the oracle's loop)
    for l in range(o.nlevs):
        ref = np.abs(off[l]).max()
        assert np.abs(forced[l] - off[l]).max() <= 1e-12 * ref, (l + 1, np.abs(forced[l] - off[l]).max() / ref)
        assert np.abs(off[l] - o.field("p", l + 1)).max() <= tol_o * ref, (l + 1, np.abs(off[l] - o.field("p", l + 1)).max() / ref)
        assert np.abs(forced[l] - o.field("p", l + 1)).max() <= tol_o * ref
        if method == "FC":
            assert np.array_equal(dflt[l], off[l]) and np.array_equal(off[l], o.field("p", l + 1)), l + 1   # four colours: the reference's bits by default
        else:
            assert np.array_equal(dflt[l], forced[l]), l + 1
    assert np.abs(forced[-1]).max() > 0
    # Vcycle(nlevs) called as an operator relaxes the p it finds (no restriction in front): the sweeps, whatever the option says
    gq = mg.grid(o.nlevs)
    pq = rng.standard_normal(gq._shape("p")); bq = rng.standard_normal(gq._shape("b"))
    mg.nhydro.set_option("coarsest_direct", 2)
    try:
        gq.set("p", pq); gq.set("b", bq); mg.fill_halo(o.nlevs, "p")
        nq = mg.nhydro.get_option("coarsest_direct_solves")
        mg.Vcycle(o.nlevs)
        assert mg.nhydro.get_option("coarsest_direct_solves") == nq
        got_q = gq.get("p")
    finally:
        mg.nhydro.set_option("coarsest_direct", 1)
    o.field("p", o.nlevs)[...] = pq; o.field("b", o.nlevs)[...] = bq; o.fill_halo(o.nlevs, "p")
    o.vcycle(o.nlevs)
    cq = o.field("p", o.nlevs)
    assert np.abs(got_q - cq).max() <= (1e-12 if method == "FC" else 1e-10) * np.abs(cq).max()
    # another matrix on the coarsest level: the operator is rebuilt (the direct solve follows the sweeps of the NEW matrix)
    gc = mg.grid(o.nlevs)
    cA = o.field("cA", o.nlevs).copy()
    cA[..., 0] *= 1.5
    gc.set("cA", cA)
    off2, _ = run(0)
    forced2, n2 = run(2)
    assert n2 == 2
    refc = np.abs(off2[-1]).max()
    assert np.abs(off2[-1] - off[-1]).max() > 1e-2 * refc     # the change is visible in the coarsest level's solution ...
    for l in range(o.nlevs):                                  # ... and the direct solve has followed it
        ref = np.abs(off2[l]).max()
        assert np.abs(forced2[l] - off2[l]).max() <= 1e-12 * ref, (l + 1, np.abs(forced2[l] - off2[l]).max() / ref)


@pytest.mark.parametrize("case", ["bmask", "tall", "stretched", "user_matrix"])
def test_rb_sequential_order_other_coefficient_paths(mg, case):
    """The sequential-order red-black (default) where the colour pass runs other kernels / other coefficients than the seamount's matrix-free
    ones: the masked system (bmask: stored coefficients, no in-kernel rebuild), nz = 128 (k_relax_tall with the snapshot), a stretched sigma
    coordinate with a moving free surface, and a matrix handed in through set_field('cA') (g = T^-1 e1 is rebuilt with the pivots).  Three
    sweeps per level from a rough random state against the oracle's sequential loop: 1e-12 of max|p|."""
    from oracle.mgoracle import Oracle, seamount_geometry
    from mgroms_amd.testcases import island_mask
    nx, ny, nz = (16, 32, 128) if case == "tall" else (32, 64, 16)
    kw = dict(relax_method="RB", bmask=(1 if case == "bmask" else 0))
    mg.nhydro_init(nx, ny, nz, 1, 1, 0, mg.nhydro.default_params(**kw))
    dx, dy, zeta, h = seamount_geometry(nx, ny, 1, 1, 0)
    hc, tb, ts = 4e3, 0.0, 0.0
    if case == "stretched":
        ii, jj = np.meshgrid(np.arange(nx + 2), np.arange(ny + 2), indexing="ij")
        zeta = 0.4 * np.cos(0.25 * ii) * np.sin(0.15 * jj); hc, tb, ts = 250.0, 0.4, 6.0
    rmask = island_mask(nx, ny) if case == "bmask" else None
    mg.nhydro_matrices(dx, dy, zeta, h, rmask, hc, tb, ts)
    o = Oracle(nx, ny, nz, 1, 1, relax_method="RB", bmask=(case == "bmask"))
    for name, a in (("dx", dx), ("dy", dy), ("zeta", zeta), ("h", h)) + ((("rmask", rmask),) if case == "bmask" else ()):
        o.field(name)[...] = a
    o.matrices(hc, tb, ts)
    if case == "user_matrix":   # the same coefficients, but through the stored-slot path of a user-supplied matrix
        for lev in range(1, o.nlevs + 1):
            mg.grid(lev).set("cA", o.field("cA", lev))
    rng = np.random.default_rng(41)
    for lev in range(1, o.nlevs + 1):
        g = mg.grid(lev)
        p = rng.standard_normal(g._shape("p")); b = rng.standard_normal(g._shape("b"))
        g.set("p", p); g.set("b", b); mg.fill_halo(lev, "p")
        o.field("p", lev)[...] = p; o.field("b", lev)[...] = b; o.fill_halo(lev, "p")
        mg.relax(lev, 3); o.relax(lev, 3)
        a, c = g.get("p"), o.field("p", lev)
        tol = 1e-12 if case != "stretched" else 1e-10   # the device's cosh / exp differ from libm's in the last bits of zr, zw (section 2)
        assert np.abs(a - c).max() <= tol * np.abs(c).max(), (case, lev, np.abs(a - c).max() / np.abs(c).max())


def test_solve_rb_sequential_order_golden(mg, golden):
    """The reference default on the reference's own recorded run (BASELINE.md 3.1, 64x64x16 on one rank, tol 1e-6): 15 iterations, every
    residual within 1e-13 + 1e-10*ref, sum(p^2) to 1e-10 -- at north_star's tolerance WITHOUT the per-plane launches of rb_exact."""
    g = golden["seamount_64x64x16_RB_1rank"]
    o = _setup(mg, 64, 64, 16, relax_method="RB", solver_prec=1e-6)
    u, v, w = _uvw(64, 64, 16)
    mg.nhydro.compute_rhs(u, v, w)
    n, hist = mg.solve_p(1e-6, 50)
    ref = np.array(g["res"])
    assert n == g["nite"] == len(ref)
    assert np.all(np.abs(hist[1:] - ref) <= 1e-13 + 1e-10 * ref), (hist[1:], ref)
    p = mg.grid(1).p
    assert np.isclose((p[1:-1, 1:-1, :] ** 2).sum(), g["sum_p2"], rtol=1e-10)
    o.field("w")[...] = w
    o.compute_rhs()
    no, ho, _ = o.solve_p()
    assert no == n and np.all(np.abs(hist - ho) <= 1e-13 + 1e-10 * np.abs(ho))
    assert np.abs(p - o.field("p")).max() <= 1e-10 * np.abs(o.field("p")).max()


def test_transfers_bitwise(mg):
    nx, ny, nz = 32, 16, 8
    for interp in ("linear", "nearest"):
        o = _setup(mg, nx, ny, nz, interp_type=interp)
        r = np.random.default_rng(3)
        g1, g2 = mg.grid(1), mg.grid(2)
        rr = r.standard_normal(g1._shape("r"))
        g1.set("r", rr); o.field("r", 1)[...] = rr
        mg.fine2coarse(1); o.fine2coarse(1)
        assert np.array_equal(g2.get("b"), o.field("b", 2))
        assert not g2.get("p").any()
        pc = r.standard_normal(g2._shape("p")); pf = r.standard_normal(g1._shape("p"))
        g2.set("p", pc); mg.fill_halo(2, "p"); g1.set("p", pf); mg.fill_halo(1, "p")
        o.field("p", 2)[...] = pc; o.fill_halo(2, "p"); o.field("p", 1)[...] = pf; o.fill_halo(1, "p")
        mg.coarse2fine(1); o.coarse2fine(1)
        assert np.array_equal(g1.get("r"), o.field("r", 1))
        assert np.array_equal(g1.get("p"), o.field("p", 1))


def test_intergrid_properties(mg):
    # the checks of the reference's unit program src/old_tests/mg_testintergrids.f90:84-128, through the C ABI and without the
    # oracle: restricting a constant gives 8x the constant, the prolongation reproduces a linear ramp away from the boundaries
    _setup(mg, 32, 32, 16)
    g1, g2 = mg.grid(1), mg.grid(2)
    g1.set("r", np.full(g1._shape("r"), 2.5))
    mg.fine2coarse(1)
    assert np.all(g2.get("b")[1:-1, 1:-1, :] == 20.0) and np.all(g2.get("p") == 0.0)
    ic, jc, kc = np.meshgrid(np.arange(18), np.arange(18), np.arange(1, 9), indexing="ij")
    g2.set("p", 3.0 * ic + 5.0 * jc + 7.0 * kc)
    g1.set("p", np.zeros(g1._shape("p")))
    mg.coarse2fine(1)
    i, j, k = np.meshgrid(np.arange(34), np.arange(34), np.arange(1, 17), indexing="ij")
    exact = 3.0 * ((i + 0.5) / 2.0) + 5.0 * ((j + 0.5) / 2.0) + 7.0 * ((k + 0.5) / 2.0)
    inner = (slice(1, 33), slice(1, 33), slice(1, 15))
    assert np.allclose(g1.get("p")[inner], exact[inner], rtol=0, atol=1e-12)


def test_compute_rhs_and_correct_uvw(mg):
    nx, ny, nz = 32, 16, 8
    o = _setup(mg, nx, ny, nz, solver_maxiter=3)
    u, v, w = _uvw(nx, ny, nz, seed=5)
    o.field("u")[...] = u; o.field("v")[...] = v; o.field("w")[...] = w
    mg.nhydro.compute_rhs(u, v, w)
    o.compute_rhs()
    assert np.array_equal(mg.grid(1).b, o.field("b"))
    mg.nhydro_solve(u, v, w)
    o.nhydro_solve()
    assert np.array_equal(mg.grid(1).p, o.field("p"))
    assert np.array_equal(u, o.field("u")) and np.array_equal(v, o.field("v")) and np.array_equal(w, o.field("w"))


@pytest.mark.parametrize("dims,geom", [((32, 32, 8), "seamount"), ((64, 32, 16), "rndtopo")])
def test_bmask_bitwise(mg, dims, geom):
    """bmask=.true.: masked coefficients on every level, masked compute_rhs / correct_uvw and the
    whole solve, bit for bit against the oracle.  The reference holds no known answers for this branch, so the
    oracle's bmask branch is itself unpinned: this test proves GPU == restatement, no more."""
    from oracle.mgoracle import Oracle, seamount_geometry, rndtopo_geometry
    from mgroms_amd.testcases import island_mask
    nx, ny, nz = dims
    kw = dict(relax_method="FC", solver_prec=1e-9, solver_maxiter=6)
    mg.nhydro_init(nx, ny, nz, 1, 1, 0, mg.nhydro.default_params(bmask=1, **kw))
    dx, dy, zeta, h = (seamount_geometry if geom == "seamount" else rndtopo_geometry)(nx, ny, 1, 1, 0)
    rmask = island_mask(nx, ny)
    mg.nhydro_matrices(dx, dy, zeta, h, rmask, 4e3, 0.0, 0.0)
    o = Oracle(nx, ny, nz, 1, 1, bmask=True, **kw)
    for name, a in (("dx", dx), ("dy", dy), ("zeta", zeta), ("h", h), ("rmask", rmask)):
        o.field(name)[...] = a
    o.matrices(4e3, 0.0, 0.0)
    for lev in range(1, o.nlevs + 1):
        g = mg.grid(lev)
        for name in ("rmask", "cw", "cA"):
            a, b = g.get(name), o.field(name, lev)
            assert np.array_equal(a, b), (lev, name, np.abs(a - b).max())
    u, v, w = _uvw(nx, ny, nz, seed=9)
    o.field("u")[...] = u; o.field("v")[...] = v; o.field("w")[...] = w
    mg.nhydro.compute_rhs(u, v, w, rmask)
    o.compute_rhs()
    assert np.array_equal(mg.grid(1).b, o.field("b"))
    mg.nhydro_solve(u, v, w, rmask)
    n, hist, _ = o.nhydro_solve()
    assert hist[-1] < 0.1 * hist[0]  # the masked system is solvable and the cycle converges on it
    assert np.array_equal(mg.grid(1).p, o.field("p"))
    assert np.array_equal(u, o.field("u")) and np.array_equal(v, o.field("v")) and np.array_equal(w, o.field("w"))
    # without rmask the masked set-up must refuse, not guess
    from mgroms_amd._lib import MgxError
    with pytest.raises(MgxError):
        mg.nhydro_matrices(dx, dy, zeta, h, None, 4e3, 0.0, 0.0)


def test_solve_fc_matches_oracle_and_golden(mg, golden):
    g = golden["seamount_64x64x16_FC_1rank"]
    o = _setup(mg, 64, 64, 16)
    u, v, w = _uvw(64, 64, 16)
    mg.nhydro.compute_rhs(u, v, w)
    n, hist = mg.solve_p(1e-10, 50)
    o.field("w")[...] = w
    o.compute_rhs()
    no, ho, _ = o.solve_p()
    assert n == no == g["nite"]
    assert np.all(np.abs(hist - ho) <= 1e-12 * np.abs(ho))   # oracle: same fields, another reduction order
    for k, ref in g["res_at"].items():  # the reference's own numbers
        assert abs(hist[int(k)] - ref) <= 1e-13 + 1e-10 * ref, (k, hist[int(k)], ref)
    p, po = mg.grid(1).p, o.field("p")
    assert np.array_equal(p, po)  # FC is bit-reproducible
    assert np.abs(p - po).max() <= 1e-10 * np.abs(po).max()


def test_solve_rb_parallel_semantics(mg, golden):
    g = golden["seamount_64x64x16_RB_1rank"]
    _setup(mg, 64, 64, 16, relax_method="RB", solver_prec=1e-6)
    u, v, w = _uvw(64, 64, 16)
    mg.nhydro.compute_rhs(u, v, w)
    mg.nhydro.set_option("rb_seq", 0)   # the plain parallel sweep: old same-colour diagonals everywhere
    try:
        n, hist = mg.solve_p(1e-6, 50)
    finally:
        mg.nhydro.set_option("rb_seq", 1)
    assert abs(n - g["nite"]) <= 1
    ref = np.array(g["res"])
    m = min(n, len(ref))
    # the reference differs from itself by 2.5e-6 between 1 and 2x2 ranks, where only the columns
    # along sub-domain edges see old same-colour diagonals; the parallel sweep sees old values everywhere: 5e-5
    assert np.all(np.abs(hist[1:m + 1] - ref[:m]) <= 5e-5 * ref[:m])
    assert np.isclose((mg.grid(1).p[1:-1, 1:-1, :] ** 2).sum(), g["sum_p2"], rtol=1e-4)


def test_16x16x8_reference_scalars(mg, golden):
    g = golden["seamount_16x16x8_FC_1rank"]
    _setup(mg, 16, 16, 8, solver_prec=1e-12)
    u, v, w = _uvw(16, 16, 8)
    mg.nhydro.compute_rhs(u, v, w)
    n, hist = mg.solve_p(1e-12, 50)
    assert n == g["nite"] and mg.nlevs() == g["nlevs"]
    p = mg.grid(1).p
    assert np.isclose((p[1:-1, 1:-1, :] ** 2).sum(), g["sum_p2"], rtol=1e-12)
    assert np.isclose(p[1, 1, 0], g["p_1_1_1"], rtol=1e-12)
    assert np.isclose(p[8, 8, 3], g["p_4_8_8"], rtol=1e-12)
    assert np.allclose(mg.grid(1).cA[8, 8, 3, :], g["cA_k4_j8_i8"], rtol=0, atol=6e-9)


def test_ragged_and_minimum_sizes(mg):
    # smallest hierarchy the reference accepts (one level) and a non-square block
    # (256,128,8): coarsest level 64x32x2 = the gathered coarsest grid of an 8-GPU run (one 1024-thread launch)
    for dims in ((4, 4, 2), (8, 16, 4), (128, 32, 8), (256, 128, 8)):
        o = _setup(mg, *dims)
        u, v, w = _uvw(*dims)
        mg.nhydro.compute_rhs(u, v, w)
        o.field("w")[...] = w
        o.compute_rhs()
        n, hist = mg.solve_p(1e-8, 5)
        no, ho, _ = o.solve_p(1e-8, 5)
        assert n == no
        assert np.array_equal(mg.grid(1).p, o.field("p")), dims


def test_stored_coefficient_path_after_set_field(mg):
    # mgx_set_field(cA) switches the smoother from the matrix-free cross terms back to the stored slots 3,5,6,8:
    # feeding the same matrix back must not change a single bit, and a modified matrix must be honoured
    nx, ny, nz = 32, 32, 16
    o = _setup(mg, nx, ny, nz)
    r = np.random.default_rng(8)
    g = mg.grid(1)
    p = r.standard_normal(g._shape("p")); b = r.standard_normal(g._shape("b"))
    cA = g.get("cA")
    for scale in (1.0, 1.5):
        cA2 = cA.copy(); cA2[..., 2] *= scale; cA2[..., 5] *= scale
        g.set("cA", cA2); o.field("cA")[...] = cA2
        g.set("p", p); g.set("b", b); mg.fill_halo(1, "p")
        o.field("p")[...] = p; o.field("b")[...] = b; o.fill_halo(1, "p")
        mg.relax(1, 1); o.relax(1, 1)
        assert np.array_equal(g.get("p"), o.field("p")), scale


def test_gauss_seidel_exact(mg):
    # relax_method='GS' (mg_relax.f90:116-148): lexicographic order reproduced exactly by hyperplane launches
    o = _setup(mg, 32, 16, 8, relax_method="GS", solver_prec=1e-8)
    r = np.random.default_rng(7)
    g = mg.grid(1)
    p = r.standard_normal(g._shape("p")); b = r.standard_normal(g._shape("b"))
    g.set("p", p); g.set("b", b); mg.fill_halo(1, "p")
    o.field("p")[...] = p; o.field("b")[...] = b; o.fill_halo(1, "p")
    mg.relax(1, 2); o.relax(1, 2)
    assert np.array_equal(g.get("p"), o.field("p"))
    u, v, w = _uvw(32, 16, 8)
    mg.nhydro.compute_rhs(u, v, w)
    o.field("w")[...] = w
    o.compute_rhs()
    n, hist = mg.solve_p(1e-8, 6)
    no, ho, _ = o.solve_p(1e-8, 6)
    assert n == no and np.array_equal(mg.grid(1).p, o.field("p"))


@pytest.mark.parametrize("method", ["FC", "RB"])
def test_tall_columns(mg, method):
    # nz=128 (BASELINE config 5): the tall-column smoother (k_relax_tall: lower 64 rows through memory, gam rebuilt on the
    # way back) and, once the matrix is user-supplied (stored slots), the generic colour pass
    o = _setup(mg, 16, 32, 128, relax_method=method, cmatrix="simple" if method == "RB" else "real")
    assert mg.grid(1).nz == 128
    u, v, w = _uvw(16, 32, 128)
    mg.nhydro.compute_rhs(u, v, w)
    o.field("w")[...] = w
    o.compute_rhs()
    n, hist = mg.solve_p(1e-8, 4)
    no, ho, _ = o.solve_p(1e-8, 4)
    assert n == no and np.array_equal(mg.grid(1).p, o.field("p"))  # RB with cmatrix='simple' is order independent too
    g = mg.grid(1)
    g.set("cA", g.get("cA"))  # same matrix, now as stored slots -> generic kernel
    r = np.random.default_rng(3)
    p = r.standard_normal(g._shape("p"))
    g.set("p", p); mg.fill_halo(1, "p")
    o.field("p")[...] = p; o.fill_halo(1, "p")
    mg.relax(1, 2); o.relax(1, 2)
    assert np.array_equal(g.get("p"), o.field("p"))


def test_stretched_sigma_coordinates(mg):
    # theta_s, theta_b > 0 exercise cosh/exp in setup_zr_zw (mg_zr_zw.f90:112-136): device libm vs host libm may
    # differ in the last bit, so this one case is compared at 1e-12 instead of bit for bit
    from oracle.mgoracle import Oracle, seamount_geometry
    nx, ny, nz = 32, 32, 16
    mg.nhydro_init(nx, ny, nz, 1, 1, 0, mg.nhydro.default_params(relax_method="FC", solver_prec=1e-9))
    dx, dy, zeta, h = seamount_geometry(nx, ny, 1, 1, 0)
    zeta = 0.3 * np.cos(np.arange(nx + 2))[:, None] * np.ones((1, ny + 2))  # non-zero free surface
    mg.nhydro_matrices(dx, dy, zeta, h, None, 250.0, 0.4, 6.0)
    o = Oracle(nx, ny, nz, relax_method="FC", solver_prec=1e-9)
    for name, a in (("dx", dx), ("dy", dy), ("zeta", zeta), ("h", h)):
        o.field(name)[...] = a
    o.matrices(250.0, 0.4, 6.0)
    for lev in range(1, o.nlevs + 1):
        for name in ("zr", "zw", "cA"):
            a, b = mg.grid(lev).get(name), o.field(name, lev)
            assert np.allclose(a, b, rtol=1e-12, atol=1e-12 * np.abs(b).max()), (lev, name)
    u, v, w = _uvw(nx, ny, nz)
    mg.nhydro.compute_rhs(u, v, w)
    o.field("w")[...] = w
    o.compute_rhs()
    n, hist = mg.solve_p(1e-9, 50)
    no, ho, _ = o.solve_p(1e-9, 50)
    # here the coefficients themselves differ in the last bits (device cosh / exp): north_star's bound, not the reduction-order one
    assert n == no and np.all(np.abs(hist - ho) <= 1e-13 + 1e-10 * np.abs(ho))


def test_device_resident_solve(mg):
    import torch
    nx, ny, nz = 32, 32, 8
    o = _setup(mg, nx, ny, nz, solver_maxiter=4)
    u, v, w = _uvw(nx, ny, nz, seed=11)
    o.field("u")[...] = u; o.field("v")[...] = v; o.field("w")[...] = w
    du, dv, dw = (torch.from_numpy(a).cuda() for a in (u, v, w))
    mg.nhydro.nhydro_solve_device(du, dv, dw)
    o.nhydro_solve()
    assert np.array_equal(du.cpu().numpy(), o.field("u")) and np.array_equal(dv.cpu().numpy(), o.field("v"))
    assert np.array_equal(dw.cpu().numpy(), o.field("w"))


def test_warm_start_and_tictoc(mg, tmp_path):
    # keep p between solves (the reference cold-starts, mg_solvers.f90:35); timer table as mg_tictoc.f90
    o = _setup(mg, 32, 32, 8)
    u, v, w = _uvw(32, 32, 8)
    mg.nhydro.compute_rhs(u, v, w)
    mg.solve_p(1e-6, 50)                 # untimed: the first launch of a kernel loads its code object (tens of ms a timer would book on that level)
    mg.nhydro.set_option("tictoc", 1)
    n1, h1 = mg.solve_p(1e-6, 50)
    n2, h2 = mg.solve_p(1e-6, 50)
    assert n1 == n2 and np.array_equal(h1, h2)          # cold start: identical repeat
    mg.nhydro.set_option("warm_start", 1)
    n3, h3 = mg.solve_p(1e-6, 50)
    assert n3 == 0 and h3[0] <= 1e-6                    # warm start: already converged
    mg.nhydro.set_option("warm_start", 0)
    f = tmp_path / "fort.10"
    mg.nhydro.print_tictoc(str(f))
    mg.nhydro.set_option("tictoc", 0)
    txt = f.read_text()
    assert "Total" in txt and "relax_3D_8_FC" in txt and "residual_3D_8" in txt and "Fcycle" in txt and "solve" in txt
    rows = [l for l in txt.splitlines() if "relax_3D_8_FC" in l]
    assert float(rows[0].split()[1]) > 0
    # the table adds up: smoothing + residuals (what the reference times inside a cycle, mg_relax.f90:128,167,209,367) are most of `solve`
    # and never more than it; no level's smoothing exceeds the whole solve (profiles/r03_tictoc_*: a level-2 entry ten times the solve's
    # share turned out to be the code-object load of an untimed-looking warm-up that the MGX_TICTOC environment switch had timed)
    tot = {l.split()[0]: [float(x) for x in l.split()[1:]] for l in txt.splitlines() if l.split() and not l.split()[0][0].isdigit() and l.split()[0] != "Total"}
    inner = tot["relax_3D_8_FC"][0] + tot["residual_3D_8"][0]
    assert 0.4 * tot["solve"][0] <= inner <= 1.02 * tot["solve"][0], tot
    assert tot["Fcycle"][0] <= 1.02 * tot["solve"][0] and max(tot["relax_3D_8_FC"][1:]) < tot["solve"][0], tot
    # byte format of print_tictoc (mg_tictoc.f90:114-153) against what the reference module itself writes (compiled unmodified with
    # flang: tests/golden/ref_tictoc.txt, oracle/make_ref_golden.py): same line shapes once digits are masked and names removed
    import os
    import re
    ref = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_tictoc.txt")).read().splitlines()
    shape = lambda l: re.sub(r"[0-9]", "#", re.sub(r"E[+-]", "E~", l))
    mine = txt.splitlines()
    nl = mg.nlevs()
    assert mine[0] == "%21s%10s" % ("", "Total") + "".join(" %9d" % l for l in range(1, nl + 1))
    assert shape(ref[0])[:41] == shape(mine[0])[:41]
    tl, cl = shape(ref[1]), shape(ref[2])   # a timer line and its call-count line, three levels wide in the fixture
    for k in range(1, len(mine), 2):
        assert shape(mine[k])[21:31] == tl[21:31] == " #.###E~##" and len(mine[k]) == 21 + 10 * (nl + 1), mine[k]
        assert re.fullmatch(r" {21}( +#+){%d}" % (nl + 1), shape(mine[k + 1])) and len(mine[k + 1]) == len(mine[k]), mine[k + 1]
        assert all(shape(mine[k])[c:c + 10] == tl[21:31] for c in range(21, len(mine[k]), 10)), mine[k]
    assert cl.startswith(" " * 21)


@pytest.mark.parametrize("dims", [(64, 64, 16), (96, 48, 16), (24, 40, 8), (256, 128, 32)])
def test_c2f_skip_is_invisible(mg, dims):
    """Inside a cycle the prolongation leaves the (i odd, j odd) columns alone when a four-colour relax follows, because that
    relax's FIRST colour is exactly (i odd, j odd) (mg_relax.f90:212-216) and overwrites them without reading them.  A/B: every
    level's p after a V-cycle and after two F-cycle iterations with the shortcut (default) and without it (option "c2f_skip" = 0,
    the reference's full update) must be the same bits -- on square, ragged and odd-quotient level shapes (24x40 -> 12x20 -> 6x10)."""
    nx, ny, nz = dims
    out = []
    for skip in (1, 0):
        mg.nhydro.set_option("c2f_skip", skip)
        try:
            _setup(mg, nx, ny, nz)
            mg.nhydro.compute_rhs(*_uvw(nx, ny, nz, seed=4))
            mg.Vcycle(1)
            pv = [mg.grid(l).p for l in range(1, mg.nlevs() + 1)]
            n, hist = mg.solve_p(1e-30, 2)
            out.append((pv, [mg.grid(l).p for l in range(1, mg.nlevs() + 1)], hist))
        finally:
            mg.nhydro.set_option("c2f_skip", 1)
    for a, b in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
        assert np.array_equal(a, b)
    assert np.array_equal(out[0][2], out[1][2])


def test_async_operators_same_bits(mg):
    """Option "async": the cycle entry points only enqueue and mgx_synchronize waits and reports.  Three V-cycles enqueued back to
    back leave the same bits as three synchronous ones, and entry points that return data to the host wait by themselves."""
    nx, ny, nz = 48, 32, 16
    out = []
    for a in (0, 1):
        mg.nhydro.set_option("async", a)
        try:
            _setup(mg, nx, ny, nz)
            mg.nhydro.compute_rhs(*_uvw(nx, ny, nz, seed=9))
            for _ in range(3):
                mg.Vcycle(1)
            if a:
                assert mg.nhydro.get_option("async") == 1
                mg.nhydro.synchronize()
            out.append([mg.grid(l).p for l in range(1, mg.nlevs() + 1)])
        finally:
            mg.nhydro.set_option("async", 0)
    for x, y in zip(*out):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("dims", [(64, 96, 32), (128, 64, 16)])
def test_transfer_kernel_variants_same_bits(dims):
    """The A/B switches of round 3's transfer kernels (read from the environment once per process, so each variant is a process of its own):
    residual+restriction by the walk on every level / without the walk on every level / four rows per wave, the walk's look-ahead depths,
    the prolongation's run length, the model kernels' run length -- the same bits as the defaults for b, every level's p and b after two
    V-cycles, after two F-cycle iterations, the history and the corrected u, v, w."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    variants = [{}, {"MGX_RESREST_FLAT_MAX": "0"}, {"MGX_RESREST_FLAT_MAX": "100000000"}, {"MGX_RESREST_FLAT_MAX": "100000000", "MGX_RESREST_FLAT_S": "4"},
                {"MGX_RESREST_FLAT_MAX": "0", "MGX_RESREST_AHEAD": "12"}, {"MGX_RESREST_FLAT_MAX": "0", "MGX_RESREST_AHEAD": "21"},
                {"MGX_C2F_KC": "1"}, {"MGX_C2F_KC": "4"}, {"MGX_MODEL_KR": "8"}, {"MGX_MODEL_KR": "1000"}]
    digests = []
    for var in variants:
        env = dict(os.environ); env.update(var)
        out = subprocess.run([sys.executable, os.path.join(here, "_gpu_variant_worker.py")] + [str(d) for d in dims], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (var, out.stdout[-2000:], out.stderr[-2000:])
        digests.append([l for l in out.stdout.splitlines() if l.startswith("DIGEST")][-1])
    for var, d in zip(variants, digests):
        assert d == digests[0], var


def test_fortran_harness_bmask(mg, tmp_path):
    """bmask = .true. through the Fortran boundary: the driver masks the boundary ring of rmask as the reference's does
    (fill_halo_2D_bmask(1,rmask) before nhydro_matrices, mg_testseamount.f90 / mg_mpi_exchange.f90:357-391; `bmask` read from the
    namelist as `use mg_namelist` gives it) and passes it to nhydro_matrices and nhydro_solve: sum(p**2), sum(div**2) and the printed
    history equal the oracle's masked solve (that oracle branch has no reference-recorded answers: DESIGN.md section 2)."""
    import os, re, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fortran", "testseamount_gpu")
    if not os.path.exists(exe):
        pytest.skip("flang not available when build() ran")
    mg.nhydro_clean()
    (tmp_path / "nh_namelist").write_text("&nhparam\n relax_method = 'FC',\n solver_prec = 1.d-9,\n bmask = .true.,\n/\n")
    out = subprocess.run([exe, "32", "48", "8"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "- bmask         : T" in out.stdout
    from oracle.mgoracle import Oracle, seamount_geometry
    nx, ny, nz = 32, 48, 8
    o = Oracle(nx, ny, nz, 1, 1, relax_method="FC", solver_prec=1e-9, bmask=True)
    for name, a in zip(("dx", "dy", "zeta", "h"), seamount_geometry(nx, ny, 1, 1, 0)):
        o.field(name)[...] = a
    m = np.ones((nx + 2, ny + 2)); m[0, :] = m[-1, :] = 0.0; m[:, 0] = m[:, -1] = 0.0
    o.field("rmask")[...] = m
    o.matrices(4e3, 0.0, 0.0)
    o.field("u")[...] = 0.0; o.field("v")[...] = 0.0
    w = o.field("w"); w[0] = 0.0; w[1:] = -1.0
    n, h, _ = o.nhydro_solve()
    its = re.findall(r"ite = *(\d+): res = *([0-9.E+-]+) / conv", out.stdout)
    assert len(its) == n
    for (k, r) in its:
        assert abs(float(r) - h[int(k)]) <= 5.1e-3 * h[int(k)]
    sp2 = float(re.search(r"sum_p2 = *([0-9.E+-]+)", out.stdout).group(1))
    assert np.isclose(sp2, (o.field("p")[1:-1, 1:-1, :] ** 2).sum(), rtol=1e-13)
    o.check_nondivergence()
    sd2 = float(re.search(r"sum_div2 = *([0-9.E+-]+)", out.stdout).group(1))
    assert np.isclose(sd2, (o.field("b")[1:-1, 1:-1, :] ** 2).sum(), rtol=1e-9)
    # and the unmasked run of the same block gives other numbers: the mask did reach the coefficients
    (tmp_path / "nh_namelist").write_text("&nhparam\n relax_method = 'FC',\n solver_prec = 1.d-9,\n/\n")
    out2 = subprocess.run([exe, "32", "48", "8"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out2.returncode == 0
    assert float(re.search(r"sum_p2 = *([0-9.E+-]+)", out2.stdout).group(1)) != sp2


@pytest.mark.parametrize("dims,par", [((128, 128, 16), {}), ((64, 128, 16), {}), ((96, 48, 16), {}), ((64, 64, 8), {}), ((128, 128, 16), {"cmatrix": "simple"}),
                                      ((48, 96, 32), {"ns_pre": 1, "ns_post": 4}), ((128, 64, 4), {}), ((256, 128, 8), {})])
def test_persistent_relax_equals_separate_launches(mg, dims, par):
    """The persistent mid-level relax (k_relax_ksp: one workgroup per plane for a whole relax call, planes handed between workgroups
    through per-plane progress counters) against one launch per colour pair (option "ksp" = 0): relax calls of 1..5 sweeps from a random
    state on every level that qualifies (level 1 here; the cycles reach them as levels 3 and 4), then two F-cycle iterations -- the
    same bits.  Run twice more with the counters already advanced (they are never reset between calls)."""
    nx, ny, nz = dims
    res = []
    for ksp in (1, 0):
        mg.nhydro.set_option("ksp", ksp)
        try:
            _setup(mg, nx, ny, nz, **par)
            rng = np.random.default_rng(11)
            got = []
            for lev in range(1, mg.nlevs() + 1):
                g = mg.grid(lev)
                if g.nz not in (4, 8, 16) or g.ny > 128 or g.nx > 128:
                    continue
                p0 = rng.standard_normal(g._shape("p")); b0 = rng.standard_normal(g._shape("b"))
                g.set("b", b0)
                for ns in (1, 2, 5, 3):
                    g.set("p", p0); mg.fill_halo(lev, "p")
                    mg.relax(lev, ns)
                    got.append(g.get("p"))
            mg.nhydro.compute_rhs(*_uvw(nx, ny, nz, seed=4))
            n, hist = mg.solve_p(1e-30, 2)
            got += [mg.grid(l).p for l in range(1, mg.nlevs() + 1)] + [hist]
            res.append(got)
        finally:
            mg.nhydro.set_option("ksp", 1)
    assert len(res[0]) == len(res[1]) > 3
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("dims,par", [((64, 64, 16), {}), ((128, 128, 16), {}), ((96, 48, 16), {}), ((64, 32, 8), {}), ((64, 64, 16), {"relax_method": "RB", "cmatrix": "simple"}),
                                      ((128, 128, 16), {"relax_method": "RB"}), ((64, 64, 16), {"ns_pre": 0, "ns_post": 1}), ((48, 96, 16), {"interp_type": "nearest"})])
def test_fused_coarse_transfers_equal_separate_operators(mg, dims, par):
    """On the level below the coarsest one the cycles fold coarse2fine in front of, and compute_residual + fine2coarse behind, the
    one-workgroup relax kernel (option "fuse_tail", default 1).  A/B against the separate operators (0): every level's p and b after a
    V-cycle from a random state, after Vcycle(nlevs-1), and after two F-cycle iterations, plus the history -- the same bits; with
    parallel red-black too (both arms run the same parallel sweep); `nearest` interpolation keeps the separate operators."""
    nx, ny, nz = dims
    res = []
    for fuse in (1, 0):
        mg.nhydro.set_option("fuse_tail", fuse)
        try:
            _setup(mg, nx, ny, nz, **par)
            nl = mg.nlevs()
            rng = np.random.default_rng(21)
            g = mg.grid(1)
            g.set("p", rng.standard_normal(g._shape("p"))); mg.fill_halo(1, "p")
            mg.nhydro.compute_rhs(*_uvw(nx, ny, nz, seed=9))
            got = []
            mg.Vcycle(1)
            got += [mg.grid(l).p for l in range(1, nl + 1)] + [mg.grid(l).b for l in range(2, nl + 1)]
            if nl >= 3:
                mg.Vcycle(nl - 1)
                got += [mg.grid(l).p for l in range(nl - 1, nl + 1)]
            n, hist = mg.solve_p(1e-30, 2)
            got += [mg.grid(l).p for l in range(1, nl + 1)] + [hist]
            c = mg.nhydro.counters()["launches"]
            res.append((got, c))
        finally:
            mg.nhydro.set_option("fuse_tail", 1)
    assert len(res[0][0]) == len(res[1][0])
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)
    if par.get("interp_type") != "nearest":
        assert res[0][1] < res[1][1]   # and it did take launches out


def test_persistent_relax_timeout_falls_back(mg):
    """The persistent relax needs all its workgroups resident together.  If one never shows up (test hook: the workgroup of plane 5
    returns at once, as if a co-tenant of the GPU kept it off the chip) its neighbours' bounded polls expire (50 ms here, 2 s in
    production), the next synchronising call fails loudly, the kernel is switched off for this instance, and the same relax call
    through one launch per colour pair gives the oracle's bits."""
    from mgroms_amd._lib import MgxError
    nx, ny, nz = 64, 64, 8
    o = _setup(mg, nx, ny, nz)
    rng = np.random.default_rng(2)
    g = mg.grid(1)
    p0 = rng.standard_normal(g._shape("p")); b0 = rng.standard_normal(g._shape("b"))
    g.set("b", b0); g.set("p", p0); mg.fill_halo(1, "p")
    o.field("p")[...] = p0; o.field("b")[...] = b0; o.fill_halo(1, "p")
    o.relax(1, 3)
    mg.nhydro.set_option("ksp_timeout_ms", 50)
    mg.nhydro.set_option("ksp_test_stall", 5)
    try:
        with pytest.raises(MgxError, match="persistent relax kernel timed out"):
            mg.relax(1, 3)
        assert mg.nhydro.get_option("ksp") == 0
        g.set("p", p0); mg.fill_halo(1, "p")
        mg.relax(1, 3)                       # separate launches now
        assert np.array_equal(g.get("p"), o.field("p"))
    finally:
        mg.nhydro.set_option("ksp_timeout_ms", 2000)
        mg.nhydro.set_option("ksp", 1)
    g.set("p", p0); mg.fill_halo(1, "p")
    mg.relax(1, 3)                           # and the persistent kernel again, counters rewound
    assert np.array_equal(g.get("p"), o.field("p"))


def test_zr_zw_kernel_against_reference_compiled_module(mg):
    """Row a13 on the device against the REAL reference: tests/golden/ref_zrzw.npz = zr, zw written by setup_zr_zw of
    mg_zr_zw.f90 compiled unmodified with flang (oracle/Makefile target `ref`).  theta = 0: no transcendental function, k_zr_zw
    must give the same bits.  theta_s, theta_b > 0: the sigma tables call cosh / exp -- the device's libm against glibc -- so
    the bound is stated in ulps of the depth scale: |d| <= 32 ulp(max|z|) (measured: a few ulp).  The level-1 colour pass rebuilds
    zw / zr in registers from the same tables; that it equals these stored arrays bit for bit is
    test_in_kernel_coefficients_match_stored_slots_on_stretched_grids (device against device)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_zrzw.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    assert len(names) >= 6
    worst = 0.0
    for n in names:
        nx, ny, nz = (int(v) for v in z[n + "/par"][:3])
        hlim, tb, ts = (float(v) for v in z[n + "/par"][3:])
        mg.nhydro_init(nx, ny, nz, 1, 1, 0, mg.nhydro.default_params(relax_method="FC"))
        dx = np.full((nx + 2, ny + 2), 100.0)
        mg.nhydro_matrices(dx, dx.copy(), z[n + "/zeta"].copy(), z[n + "/h"].copy(), None, hlim, tb, ts)
        ring = (slice(1, -1), slice(1, -1))
        for name in ("zr", "zw"):
            a, g = mg.grid(1).get(name)[ring], z[n + "/" + name][ring]
            if tb == 0.0 and ts == 0.0:
                assert np.array_equal(a, g), (n, name)
            else:
                ulp = np.spacing(np.abs(g).max())
                d = np.abs(a - g).max() / ulp
                worst = max(worst, d)
                assert d <= 32, (n, name, d)
    print(f"zr/zw with cosh/exp tables: max deviation {worst:.1f} ulp of the depth scale")


def test_error_behaviour(mg):
    from mgroms_amd._lib import MgxError
    with pytest.raises(MgxError):
        mg.nhydro_init(15, 16, 8, 1, 1, 0, mg.nhydro.default_params())  # odd size
    with pytest.raises(MgxError):
        mg.nhydro_init(16, 16, 8, 1, 1, 0, mg.nhydro.default_params(interp_type="linear", restrict_type="linear"))
    with pytest.raises(MgxError):
        mg.nhydro_init(16, 16, 8, 3, 1, 0, mg.nhydro.default_params())  # process grid not a power of two
    with pytest.raises(MgxError):
        mg.nhydro_init(20, 16, 16, 1, 1, 0, mg.nhydro.default_params())  # 20 -> 10 -> 5: an odd local size on level 3
    mg.nhydro_init(16, 16, 8, 1, 1, 0, mg.nhydro.default_params())
    with pytest.raises(MgxError):
        mg.relax(9, 1)
    with pytest.raises(MgxError):
        mg.solve_p(1e-6, 5)  # no matrices yet


def test_full_size_properties(mg, golden):
    """BASELINE config 3 (512x512x64, FC): first five residuals against the reference's printed digits, plus
    size-independent properties: restriction of a constant, and linearity of the residual in p."""
    nx, ny, nz = 512, 512, 64
    from oracle.mgoracle import seamount_geometry
    mg.nhydro_init(nx, ny, nz, 1, 1, 0, mg.nhydro.default_params(relax_method="FC", solver_maxiter=5))
    dx, dy, zeta, h = seamount_geometry(nx, ny, 1, 1, 0)
    mg.nhydro_matrices(dx, dy, zeta, h, None, 4e3, 0.0, 0.0)
    u, v, w = _uvw(nx, ny, nz)
    mg.nhydro.compute_rhs(u, v, w)
    n, hist = mg.solve_p(1e-12, 5)
    ref = golden["seamount_512x512x64_FC_first5_printed"]
    assert np.all(np.abs(hist[1:6] - ref) <= 6e-4), hist
    g1 = mg.grid(1)
    g1.set("r", np.ones(g1._shape("r")))
    mg.fine2coarse(1)
    assert np.all(mg.grid(2).b[1:-1, 1:-1, :] == 8.0)
    rng = np.random.default_rng(0)
    sh = g1._shape("p")
    pa = rng.standard_normal(sh); pb = rng.standard_normal(sh)
    g1.set("b", np.zeros(sh))
    out = []
    for q in (pa, pb, pa + pb):
        g1.set("p", q); mg.fill_halo(1, "p"); mg.compute_residual(1); out.append(g1.get("r")[1:-1, 1:-1, :])
    assert np.abs(out[2] - out[0] - out[1]).max() <= 1e-9 * np.abs(out[2]).max()


def test_vcycle2_and_fcycle_ops(mg):
    # Vcycle2(lev1,lev2) (mg_solvers.f90:155-177) with lev2 = nlevs is Vcycle(lev1); Fcycle against the oracle
    o = _setup(mg, 32, 32, 16)
    u, v, w = _uvw(32, 32, 16)
    mg.nhydro.compute_rhs(u, v, w)
    o.field("w")[...] = w
    o.compute_rhs()
    mg.Vcycle2(1, mg.nlevs()); o.vcycle(1)
    assert np.array_equal(mg.grid(1).p, o.field("p"))
    # Fcycle starts by restricting grid(1)%r (mg_solvers.f90:110-114): solve_p computes the residual first (:50), so does this test
    mg.compute_residual(1); o.residual(1)
    mg.Fcycle(); o.fcycle()
    assert np.array_equal(mg.grid(1).p, o.field("p"))
    for lev in range(2, o.nlevs + 1):
        assert np.array_equal(mg.grid(lev).p, o.field("p", lev)), lev


@pytest.mark.parametrize("dims", [(128, 128, 32), (64, 128, 16), (256, 256, 64), (32, 32, 8)])
def test_fcycle_first_leg_as_one_launch(mg, dims):
    """Fcycle's first leg below level 1 (mg_solvers.f90:110-115: restrict r, r_c = b_c, p_c = 0, level after level) runs as ONE launch that carries a
    block of the finest level of the chain down through LDS (k_restrict_chain; up to four levels at a time).  A/B against one launch per level
    (option "restrict_chain" = 0): b, r and p of every level right after an F-cycle, and the solve's history -- the same bits; against the
    oracle too; and it did take launches out."""
    nx, ny, nz = dims
    res = []
    for chain in (1, 0):
        mg.nhydro.set_option("restrict_chain", chain)
        try:
            o = _setup(mg, nx, ny, nz)
            u, v, w = _uvw(nx, ny, nz, seed=3)
            mg.nhydro.compute_rhs(u, v, w)
            mg.compute_residual(1)
            c0 = mg.nhydro.counters()["launches"]
            mg.Fcycle()
            c1 = mg.nhydro.counters()["launches"]
            got = [mg.grid(l).get(f) for l in range(1, mg.nlevs() + 1) for f in ("p", "b")]
            n, hist = mg.solve_p(1e-30, 2)
            got += [mg.grid(1).p, hist]
            res.append((got, c1 - c0))
            if chain:
                o.field("u")[...] = u; o.field("v")[...] = v; o.field("w")[...] = w
                o.compute_rhs(); o.residual(1); o.fcycle()
                for l in range(1, o.nlevs + 1):
                    assert np.array_equal(got[2 * (l - 1)], o.field("p", l)) and np.array_equal(got[2 * (l - 1) + 1], o.field("b", l)), l
        finally:
            mg.nhydro.set_option("restrict_chain", 1)
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)
    if mg.nlevs() >= 4:
        assert res[0][1] < res[1][1]


def test_cycle_keeps_the_dead_r_on_request(mg):
    # coarse2fine leaves the interpolated correction in the fine r (mg_intergrids.f90:218-226); nothing reads it before the next
    # compute_residual, so the cycles skip that store unless "keep_r" (or "exact_halos") asks for the reference's state
    o = _setup(mg, 32, 32, 16)
    u, v, w = _uvw(32, 32, 16)
    mg.nhydro.compute_rhs(u, v, w)
    o.field("w")[...] = w
    o.compute_rhs()
    mg.nhydro.set_option("keep_r", 1)
    try:
        mg.Vcycle(1); o.vcycle(1)
        for lev in range(1, o.nlevs):
            assert np.array_equal(mg.grid(lev).p, o.field("p", lev)), lev
            assert np.array_equal(mg.grid(lev).r[1:-1, 1:-1, :], o.field("r", lev)[1:-1, 1:-1, :]), lev
    finally:
        mg.nhydro.set_option("keep_r", 0)
    mg.Vcycle(1); o.vcycle(1)
    for lev in range(1, o.nlevs + 1):
        assert np.array_equal(mg.grid(lev).p, o.field("p", lev)), lev


def test_constant_divisor_quotient_is_the_hardware_quotient(mg):
    # The colour pass divides by per-column constants through a reciprocal refined with the two Newton steps of the hardware fp64
    # division sequence and then that sequence's last three operations (DIVC, mgx_device.h).  Same operations on the same values as
    # `/` while nothing needs rescaling: bit-identical for every pair below (magnitudes far beyond any depth or metric factor, zero and
    # negative numerators included).
    import ctypes as C
    from mgroms_amd._lib import lib, check
    r = np.random.default_rng(7)
    n = 1 << 20
    a = r.standard_normal(n) * 10.0 ** r.uniform(-100, 100, n)
    b = (r.uniform(0.5, 2.0, n) * 10.0 ** r.uniform(-100, 100, n)) * np.where(r.random(n) < 0.5, 1.0, -1.0)
    a[:1000] = 0.0
    a[1000:2000] = b[1000:2000]          # quotient exactly one
    b[2000:3000] = 2.0 ** r.integers(-50, 50, 1000)   # powers of two
    nbad = C.c_longlong(-1)
    check(lib().mgx_selftest_divc(a.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)), n, C.byref(nbad)))
    assert nbad.value == 0


def test_fortran_harness(mg, tmp_path):
    """The reference's driver shape in Fortran (fortran/mg_testseamount_gpu.f90) over module nhydro -> ISO_C_BINDING
    -> libmgx.so: same residual history as the oracle, printed in the reference's format."""
    import os, re, shutil, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fortran", "testseamount_gpu")
    if not os.path.exists(exe):
        pytest.skip("flang not available when build() ran")
    mg.nhydro_clean()
    (tmp_path / "nh_namelist").write_text("&nhparam\n relax_method = 'FC',\n solver_prec = 1.d-10,\n/\n")
    out = subprocess.run([exe, "64", "64", "16"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    its = re.findall(r"ite = *(\d+): res = *([0-9.E+-]+) / conv", out.stdout)
    assert len(its) == 25
    from oracle.mgoracle import make_seamount
    o = make_seamount(64, 64, 16, relax_method="FC", solver_prec=1e-10)
    n, h, _ = o.nhydro_solve()
    for (k, r) in its:
        assert abs(float(r) - h[int(k)]) <= 5.1e-3 * h[int(k)]  # Fortran E10.3: 0.dddE+ee, three significant digits
    sp2 = float(re.search(r"sum_p2 = *([0-9.E+-]+)", out.stdout).group(1))
    # p is taken after correct_uvw, unchanged by it
    assert np.isclose(sp2, (o.field("p")[1:-1, 1:-1, :] ** 2).sum(), rtol=1e-13)
    o.check_nondivergence()
    sd2 = float(re.search(r"sum_div2 = *([0-9.E+-]+)", out.stdout).group(1))
    assert np.isclose(sd2, (o.field("b")[1:-1, 1:-1, :] ** 2).sum(), rtol=1e-9)
    assert os.path.exists(tmp_path / "fort.100")  # convergence history file (mg_solvers.f90:59,72)
    # the summary block of solve_p (mg_solvers.f90:93-96) and Fortran's E10.3 in the history lines (format 10, :99)
    assert " --- summary ---" in out.stdout and "time spent to solve :" in out.stdout and "rescaled performance:" in out.stdout
    assert re.search(r"ite =  1: res =  0\.\d{3}E[+-]\d{2} / conv = ", out.stdout), out.stdout[-1500:]


def test_fortran_solver_surface(mg, tmp_path):
    """The solver-level half of the Fortran boundary (fortran/mg_solvers.f90, re-exported by module nhydro as the reference's `use` chain
    does): relax / compute_residual / fill_halo per level, fine2coarse / coarse2fine, Vcycle, Vcycle2, Fcycle, solve_p, grid_get / grid_set,
    nlevs / myrank / netcdf_output, tic / toc / print_tictoc -- driven by fortran/mg_testrelax_gpu.f90 (the shape of the reference's
    old_tests/mg_testrelax.f90) through flang -> ISO_C_BINDING -> libmgx.so, every printed number against the same sequence on the oracle
    (fields are bit-identical with FC; the printed sums and norms agree to 1e-12)."""
    import os, re, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fortran", "testrelax_gpu")
    if not os.path.exists(exe):
        pytest.skip("flang not available when build() ran")
    mg.nhydro_clean()
    nx, ny, nz = 64, 32, 16
    (tmp_path / "nh_namelist").write_text("&nhparam\n relax_method = 'FC',\n netcdf_output = .false.,\n/\n")
    out = subprocess.run([exe, str(nx), str(ny), str(nz)], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    txt = out.stdout
    from oracle.mgoracle import make_seamount
    o = make_seamount(nx, ny, nz, relax_method="FC")
    assert re.search(r"nlevs = *%d netcdf_output = *F myrank = *0" % o.nlevs, txt), txt[:600]

    def num(pat):
        m = re.search(pat + r" *([0-9.E+-]+)", txt)
        assert m, (pat, txt[-2000:])
        return float(m.group(1))

    def close(a, b):
        return abs(a - b) <= 1e-12 * abs(b)

    for lev in range(1, o.nlevs + 1):
        li = o.level_info(lev)
        i, j, k = np.meshgrid(np.arange(1, li["nx"] + 1), np.arange(1, li["ny"] + 1), np.arange(1, li["nz"] + 1), indexing="ij")
        p = o.field("p", lev); b = o.field("b", lev)
        p[...] = 0.0; b[...] = 0.0
        p[1:-1, 1:-1, :] = ((7 * k + 3 * j + 5 * i + lev) % 11) / 11.0 - 0.5
        b[1:-1, 1:-1, :] = ((5 * k + 7 * j + 3 * i + 2 * lev) % 13) / 13.0 - 0.5
        o.fill_halo(lev, "p")
        o.relax(lev, 2)
        res = o.residual(lev)
        m = re.search(r"lev= *%d res= *([0-9.E+-]+) sum_p2= *([0-9.E+-]+)" % lev, txt)
        assert m, (lev, txt[-2000:])
        assert close(float(m.group(1)), res), (lev, m.group(1), res)
        assert close(float(m.group(2)), (o.field("p", lev)[1:-1, 1:-1, :] ** 2).sum()), lev
    o.fine2coarse(1)
    assert close(num("f2c_sum_b2 ="), (o.field("b", 2)[1:-1, 1:-1, :] ** 2).sum())
    o.coarse2fine(1)
    assert close(num("c2f_sum_p2 ="), (o.field("p", 1)[1:-1, 1:-1, :] ** 2).sum())
    o.vcycle(1)
    assert close(num("vcycle_res ="), o.residual(1))
    # Vcycle2(1,3) (mg_solvers.f90:155-177) spelled out with the oracle's operators: ns_pre = 3, ns_post = 2, ns_coarsest = 40
    o.relax(1, 3); o.residual(1); o.fine2coarse(1)
    o.relax(2, 3); o.residual(2); o.fine2coarse(2)
    o.relax(3, 40)
    o.coarse2fine(2); o.relax(2, 2)
    o.coarse2fine(1); o.relax(1, 2)
    assert close(num("vcycle2_res ="), o.residual(1))
    o.fcycle()
    assert close(num("fcycle_res ="), o.residual(1))
    u, v, w = _uvw(nx, ny, nz)
    o.field("u")[...] = u; o.field("v")[...] = v; o.field("w")[...] = w
    o.compute_rhs()
    n, hist, _ = o.solve_p(1e-8, 5)
    m = re.search(r"solve_p_nite = *(\d+) res = *([0-9.E+-]+) sum_p2 = *([0-9.E+-]+)", txt)
    assert m and int(m.group(1)) == n == 5
    assert close(float(m.group(2)), hist[-1]) and close(float(m.group(3)), (o.field("p")[1:-1, 1:-1, :] ** 2).sum())
    # tic / toc / print_tictoc of the caller's own section: the reference's table shape in fort.10, with the program's timer in it
    tt = (tmp_path / "fort.10").read_text()
    assert "Total" in tt.splitlines()[0] and re.search(r"^ +mg_testrelax 0\.\d{3}E[+-]\d{2} 0\.\d{3}E[+-]\d{2}$", tt, re.M), tt
