"""BASELINE.json configurations at their own sizes, through the C ABI on the GPU, against the CPU oracle (and the reference's
printed digits, tests/golden): config 2 (seamount 256x256x32, red-black), config 3 (512x512x64, four-colour, one V-cycle bit
for bit), config 4 (rndtopo 1024x1024x64 on one GPU against 4x4 emulated ranks; the 2x2 GPU decomposition of the same
generator is in test_gpu_multirank.py), and the exact-order red-black mode that makes the reference default
(relax_method='RB', cmatrix='real') a bit-for-bit case.

Tolerances: FC, and RB with rb_exact=1: fields compared with np.array_equal; residual histories |d| <= 1e-13 + 1e-10*ref in
units of ||b|| (norms are reduced in a different order).  Parallel (default) RB: 5e-5 relative on the history -- the sweep
reads the same-colour k=1 diagonals as they were before the pass, the reference's sequential loop sees half of them updated
(the reference differs from ITSELF by 2.5e-6 between 1 and 2x2 ranks for the same reason)."""
import os

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
    m.nhydro.set_option("rb_exact", 0)
    m.nhydro_clean()


def _gpu(mg, nx, ny, nz, geom="seamount", **par):
    from mgroms_amd.testcases import seamount_geometry, rndtopo_geometry, resting_column_state
    mg.nhydro_init(nx, ny, nz, 1, 1, 0, mg.nhydro.default_params(**par))
    dx, dy, zeta, h = (seamount_geometry if geom == "seamount" else rndtopo_geometry)(nx, ny)
    mg.nhydro_matrices(dx, dy, zeta, h, None, 4e3, 0.0, 0.0)
    u, v, w = resting_column_state(nx, ny, nz)
    mg.nhydro.compute_rhs(u, v, w)


def _oracle(nx, ny, nz, npx=1, npy=1, geom="seamount", threads=None, **par):
    """the same problem on npx x npy emulated ranks (one OpenMP thread each)"""
    from oracle.mgoracle import Oracle, seamount_geometry, rndtopo_geometry
    o = Oracle(nx // npx, ny // npy, nz, npx, npy, **par)
    g = seamount_geometry if geom == "seamount" else rndtopo_geometry
    for r in range(npx * npy):
        for name, a in zip(("dx", "dy", "zeta", "h"), g(nx // npx, ny // npy, npx, npy, r)):
            o.field(name, 1, r)[...] = a
    o.matrices(4e3, 0.0, 0.0)
    for r in range(npx * npy):
        o.field("u", 1, r)[...] = 0.0
        o.field("v", 1, r)[...] = 0.0
        w = o.field("w", 1, r)
        w[0] = 0.0
        w[1:] = -1.0
    o.compute_rhs()
    return o


def _blocks_equal(p, o, name, npx, npy):
    """the one-rank GPU field against the oracle's per-rank blocks (interiors)"""
    lx, ly = (p.shape[0] - 2) // npx, (p.shape[1] - 2) // npy
    bad = []
    for r in range(npx * npy):
        pi, pj = r % npx, r // npx
        blk = o.field(name, 1, r)[1:-1, 1:-1, :]
        mine = p[1 + pi * lx:1 + (pi + 1) * lx, 1 + pj * ly:1 + (pj + 1) * ly, :]
        if not np.array_equal(blk, mine):
            bad.append((r, float(np.abs(blk - mine).max())))
    return bad


def _hist_close(h, ho, rtol=1e-12):
    """against the ORACLE: the fields are bit-identical, only the order of the norm's reduction differs -> 1e-12 relative.
    Against the reference's recorded series (tests/golden, another build's libm and reduction order): rtol=1e-10 + 1e-13, north_star's bound."""
    return np.all(np.abs(h - ho) <= (1e-13 if rtol > 1e-11 else 0.0) + rtol * np.abs(ho))


# ---- exact-order red-black -----------------------------------------------------------------------------------------
def test_rb_exact_matches_reference_history_and_oracle(mg, golden):
    """relax_method='RB', cmatrix='real' (the reference default) in the reference's sequential order: the 15 residuals of
    BASELINE.md 3.1 (1 rank) within 1e-13 + 1e-10*ref, p bit for bit against the oracle."""
    g = golden["seamount_64x64x16_RB_1rank"]
    mg.nhydro.set_option("rb_exact", 1)
    try:
        _gpu(mg, 64, 64, 16, relax_method="RB", solver_prec=1e-6)
        n, hist = mg.solve_p(1e-6, 50)
    finally:
        mg.nhydro.set_option("rb_exact", 0)
    ref = np.array(g["res"])
    assert n == g["nite"] == len(ref)
    assert _hist_close(hist[1:], ref, rtol=1e-10), (hist[1:], ref)
    o = _oracle(64, 64, 16, relax_method="RB", solver_prec=1e-6)
    no, ho, _ = o.solve_p(1e-6, 50)
    assert no == n and _hist_close(hist, ho)
    p = mg.grid(1).p
    assert np.array_equal(p, o.field("p"))
    assert np.isclose((p[1:-1, 1:-1, :] ** 2).sum(), g["sum_p2"], rtol=1e-12)
    # the plain parallel sweep on the same problem differs, within the stated tolerance
    _gpu(mg, 64, 64, 16, relax_method="RB", solver_prec=1e-6)
    mg.nhydro.set_option("rb_seq", 0)
    try:
        n2, hist2 = mg.solve_p(1e-6, 50)
    finally:
        mg.nhydro.set_option("rb_seq", 1)
    assert n2 == n and not np.array_equal(hist2, hist) and np.all(np.abs(hist2[1:] - ref) <= 5e-5 * ref)
    # and the default (sequential order at speed, mgx_rbseq.hip) is the exact one up to a few ulp
    _gpu(mg, 64, 64, 16, relax_method="RB", solver_prec=1e-6)
    n3, hist3 = mg.solve_p(1e-6, 50)
    assert n3 == n and _hist_close(hist3, hist, rtol=1e-10)
    assert np.abs(mg.grid(1).p - p).max() <= 1e-10 * np.abs(p).max()


@pytest.mark.parametrize("dims", [(32, 16, 8), (16, 64, 4), (128, 32, 8)])
def test_rb_exact_relax_every_level(mg, dims):
    """one relax call per level (per-plane launches on the large levels, the one-workgroup kernels on the small ones) from a
    random state: bit-identical to the oracle's sequential loop; ragged blocks included"""
    nx, ny, nz = dims
    mg.nhydro.set_option("rb_exact", 1)
    try:
        _gpu(mg, nx, ny, nz, relax_method="RB")
        o = _oracle(nx, ny, nz, relax_method="RB")
        rng = np.random.default_rng(5)
        for lev in range(1, o.nlevs + 1):
            g = mg.grid(lev)
            p = rng.standard_normal(g._shape("p")); b = rng.standard_normal(g._shape("b"))
            g.set("p", p); g.set("b", b); mg.fill_halo(lev, "p")
            o.field("p", lev)[...] = p; o.field("b", lev)[...] = b; o.fill_halo(lev, "p")
            mg.relax(lev, 2); o.relax(lev, 2)
            assert np.array_equal(g.get("p"), o.field("p", lev)), lev
    finally:
        mg.nhydro.set_option("rb_exact", 0)


def test_relax_rb_parallel_deviation_is_small(mg):
    """What the parallel red-black sweep (default) actually differs by from the sequential one after ONE sweep on a smooth
    field (the solve's own iterate): the k=1 diagonal coupling times the change of one sweep."""
    _gpu(mg, 64, 64, 16, relax_method="RB")
    mg.Vcycle(1)
    p0, b0 = mg.grid(1).p, mg.grid(1).b
    o = _oracle(64, 64, 16, relax_method="RB")
    o.field("p")[...] = p0
    o.field("b")[...] = b0
    mg.nhydro.set_option("rb_seq", 0)
    try:
        mg.relax(1, 1); o.relax(1, 1)
    finally:
        mg.nhydro.set_option("rb_seq", 1)
    a, c = mg.grid(1).p, o.field("p")
    d = np.abs(a - c).max() / np.abs(c).max()
    assert 0 < d <= 1e-4, d
    # the default sweep from the same state: the sequential order itself
    mg.grid(1).set("p", p0); mg.fill_halo(1, "p")
    mg.relax(1, 1)
    assert np.abs(mg.grid(1).p - c).max() <= 1e-13 * np.abs(c).max()


# ---- BASELINE config 2: seamount 256x256x32, red-black ---------------------------------------------------------------
def test_config2_rb_256x256x32(mg, golden):
    ref = golden["seamount_256x256x32_RB_printed"]
    # (a) the default (sequential order at speed), 50 iterations as the reference runs it (it stops at maxiter)
    _gpu(mg, 256, 256, 32, relax_method="RB", solver_prec=1e-8)
    n, hist = mg.solve_p(1e-8, 50)
    assert n == 50 == ref["nite"]
    assert np.all(np.abs(hist[1:6] - np.array(ref["first5"])) <= 6e-4), hist[1:6]       # printed with 3 digits
    assert abs(hist[50] - ref["res50"]) <= 0.0006e-5, hist[50]                          # 0.426E-05
    hist_seq = hist.copy()
    # (a') the plain parallel sweep: the same printed digits, 5e-5 away
    _gpu(mg, 256, 256, 32, relax_method="RB", solver_prec=1e-8)
    mg.nhydro.set_option("rb_seq", 0)
    try:
        n, hist = mg.solve_p(1e-8, 50)
    finally:
        mg.nhydro.set_option("rb_seq", 1)
    assert n == 50 and abs(hist[50] - ref["res50"]) <= 0.0006e-5 + 5e-5 * ref["res50"], hist[50]
    hist_par = hist.copy()
    # (b) exact order: first 3 iterations bit for bit against the oracle on one rank; the parallel sweep within 5e-5 of it
    o = _oracle(256, 256, 32, relax_method="RB")
    no, ho, _ = o.solve_p(1e-12, 3)
    mg.nhydro.set_option("rb_exact", 1)
    try:
        _gpu(mg, 256, 256, 32, relax_method="RB")
        n, hist = mg.solve_p(1e-12, 3)
    finally:
        mg.nhydro.set_option("rb_exact", 0)
    assert n == no == 3 and _hist_close(hist, ho)
    assert np.array_equal(mg.grid(1).p, o.field("p"))
    assert np.all(np.abs(hist_par[1:4] - ho[1:]) <= 5e-5 * ho[1:])
    assert np.all(np.abs(hist_seq[1:4] - ho[1:]) <= 1e-13 + 1e-10 * ho[1:]), (hist_seq[1:4], ho[1:])
    assert np.all(np.abs(ho[1:4] - np.array(ref["first5"][:3])) <= 6e-4)


def test_rb_sequential_order_512x512x64_against_the_exact_order(mg):
    """BASELINE config 3's size (the bench's workload) with the reference DEFAULT ordering, relax_method='RB', cmatrix='real': two solve_p
    iterations in the sequential order at speed (rb_seq -- at this size the level-1 correction runs inside the walk's launch across the XCDs,
    levels 3-4 walk per workgroup, the coarsest level walks in registers; and, the default since, the windowed walk on levels 1-4) against the same two iterations in the bit-exact order (rb_exact, one
    launch per plane: the mode the smaller tests pin on the oracle, which would need minutes here): p within 1e-12 of max|p|, residuals within
    1e-13 + 1e-10 * ref -- north_star's bound -- and the fused launches really ran (fewer launches than with rbseq_fuse = 0, same bits)."""
    nx, ny, nz = 512, 512, 64
    mg.nhydro.set_option("rb_exact", 1)
    try:
        _gpu(mg, nx, ny, nz, relax_method="RB")
        ne, he = mg.solve_p(1e-30, 2)
        pe = mg.grid(1).p.copy()
    finally:
        mg.nhydro.set_option("rb_exact", 0)
    got = {}
    for mode, window, fuse in (("window", 1, 1), (1, 0, 1), (0, 0, 0)):   # the default (windowed walk), then the walk over the whole level: fused launch, separate launches
        mg.nhydro.set_option("rbseq_fuse", fuse); mg.nhydro.set_option("rbseq_window", window)
        try:
            _gpu(mg, nx, ny, nz, relax_method="RB")
            n0 = mg.nhydro.counters()["launches"]; w0 = mg.nhydro.get_option("rbseq_window_colours")
            n, h = mg.solve_p(1e-30, 2)
            got[mode] = (mg.grid(1).p.copy(), h.copy(), mg.nhydro.counters()["launches"] - n0, mg.nhydro.get_option("rbseq_window_colours") - w0)
            assert mg.nhydro.get_option("rbseq_fuse") == fuse    # (a lost hand-off would have switched it off -- and failed the call)
        finally:
            mg.nhydro.set_option("rbseq_fuse", 1); mg.nhydro.set_option("rbseq_window", 1)
    assert n == ne == 2
    for mode in ("window", 1):
        p, h, launches, wcol = got[mode]
        assert np.abs(p - pe).max() <= 1e-12 * np.abs(pe).max(), (mode, np.abs(p - pe).max() / np.abs(pe).max())
        assert np.all(np.abs(h[1:] - he[1:]) <= 1e-13 + 1e-10 * he[1:]), (mode, h, he)
    assert np.array_equal(got[1][0], got[0][0]) and np.array_equal(got[1][1], got[0][1])
    assert got[1][2] < got[0][2], (got[1][2], got[0][2])
    # the windowed walk served every colour of levels 1-4 (levels 5, 6 run the plane loop inside their one-workgroup kernels), no launch more than the fused walk
    assert got["window"][3] > 0 and got[1][3] == 0 and got[0][3] == 0 and got["window"][2] <= got[1][2], [g[2:] for g in got.values()]


def test_config2_fc_256x256x32_bitwise(mg):
    _gpu(mg, 256, 256, 32, relax_method="FC")
    n, hist = mg.solve_p(1e-12, 3)
    o = _oracle(256, 256, 32, 4, 2, relax_method="FC")
    no, ho, _ = o.solve_p(1e-12, 3)
    assert n == no == 3 and _hist_close(hist, ho)
    assert _blocks_equal(mg.grid(1).p, o, "p", 4, 2) == []


# ---- BASELINE config 3: seamount 512x512x64, four-colour, one V-cycle bit for bit ---------------------------------------
@pytest.mark.parametrize("dims,geom", [((96, 48, 16), "seamount"), ((48, 96, 32), "rndtopo"), ((64, 128, 8), "seamount"), ((128, 64, 16), "rndtopo")])
def test_fc_ragged_shapes_bitwise(mg, dims, geom):
    # shapes whose levels exercise the partly filled waves of the mid- and coarse-level kernels (a plane of 24 / 48 / 64 columns per
    # colour for the colour-pair kernel, 72- and 18-block levels for the register-resident coarse kernels, runs of the prolongation
    # that end inside a wave): three F-cycle iterations, every level's p bit for bit
    nx, ny, nz = dims
    _gpu(mg, nx, ny, nz, geom, relax_method="FC")
    o = _oracle(nx, ny, nz, geom=geom, relax_method="FC")
    assert mg.nlevs() == o.nlevs
    n, hist = mg.solve_p(1e-30, 3)
    no, ho, _ = o.solve_p(1e-30, 3)
    assert n == no == 3 and _hist_close(hist, ho)
    for lev in range(1, o.nlevs + 1):
        assert np.array_equal(mg.grid(lev).p, o.field("p", lev)), lev


def test_config3_vcycle_512x512x64_bitwise(mg, golden):
    _gpu(mg, 512, 512, 64, relax_method="FC")
    b_gpu = mg.grid(1).b
    mg.Vcycle(1)
    res = mg.compute_residual(1)
    o = _oracle(512, 512, 64, 4, 4, relax_method="FC")
    assert _blocks_equal(b_gpu, o, "b", 4, 4) == []
    o.vcycle(1)
    reso = o.residual(1)
    assert _blocks_equal(mg.grid(1).p, o, "p", 4, 4) == []
    assert _blocks_equal(mg.grid(1).r, o, "r", 4, 4) == []
    assert abs(res - reso) <= 1e-12 * reso
    o.close()


# ---- BASELINE config 5: seamount 2048x2048x128 on 4x2 -> the local block 512x1024x128 of one GPU ------------------------------
def test_config5_vcycle_512x1024x128_bitwise(mg):
    """Config 5's per-GPU block (2048/4 x 2048/2 x 128) as a one-rank problem: nx /= ny, nz = 128 (k_relax_tall at full plane
    width, the 7-level hierarchy 512x1024x128 -> 8x16x2 of SURVEY 8(a14)), one FC Vcycle(1) bit for bit (b, p, r) against the
    oracle on 4x4 emulated ranks, like config 3.  The 4x2 process grid itself, with the gather chain, is
    tests/test_gpu_multirank.py::test_config5_4x2_ranks_in_one_process."""
    _gpu(mg, 512, 1024, 128, relax_method="FC")
    assert mg.nlevs() == 7
    b_gpu = mg.grid(1).b
    mg.Vcycle(1)
    res = mg.compute_residual(1)
    p, r = mg.grid(1).p, mg.grid(1).r
    mg.nhydro_clean()
    o = _oracle(512, 1024, 128, 4, 4, relax_method="FC")
    assert o.nlevs == 7
    assert _blocks_equal(b_gpu, o, "b", 4, 4) == []
    o.vcycle(1)
    reso = o.residual(1)
    assert _blocks_equal(p, o, "p", 4, 4) == []
    assert _blocks_equal(r, o, "r", 4, 4) == []
    assert abs(res - reso) <= 1e-12 * reso
    o.close()


def test_coarse_levels_of_an_eight_gpu_run_bitwise(mg):
    """What eight GPUs (4x2 ranks of 512x512x64, nsmall = 256) run redundantly after the gathers: the global coarse levels 256x128x8,
    128x64x4 and the 64x32x2 coarsest grid with its 40 sweeps.  As a one-rank problem (256x128x8 -> 3 levels) every kernel that serves them
    -- the colour-pair kernel (nx = 256 is beyond the persistent one), the persistent relax at nz = 4, the eight-wave one-workgroup
    coarsest solve -- bit for bit against the oracle over three F-cycle iterations, every level's p."""
    _gpu(mg, 256, 128, 8, relax_method="FC")
    o = _oracle(256, 128, 8, relax_method="FC")
    assert mg.nlevs() == o.nlevs == 3
    n, hist = mg.solve_p(1e-30, 3)
    no, ho, _ = o.solve_p(1e-30, 3)
    assert n == no == 3 and _hist_close(hist, ho)
    for lev in range(1, 4):
        assert np.array_equal(mg.grid(lev).p, o.field("p", lev)), lev
    g = mg.grid(3)
    assert (g.nx, g.ny, g.nz) == (64, 32, 2)


@pytest.mark.parametrize("nz", [64, 128])
def test_level1_red_black_two_waves_per_block(mg, nz):
    """The level-1 kernels with TWO waves per workgroup (512 planes x 4 j-chunks = 2048 waves: red-black at 512x512; the LDS slices of
    gam / of the lower rows are per wave): two sweeps from p = 0, bit for bit against the oracle on 4x4 emulated ranks.
    cmatrix='simple' makes red-black order independent, so the parallel sweep is exact."""
    _gpu(mg, 512, 512, nz, relax_method="RB", cmatrix="simple")
    mg.relax(1, 2)
    p = mg.grid(1).p
    mg.nhydro_clean()
    o = _oracle(512, 512, nz, 4, 4, relax_method="RB", cmatrix="simple")
    o.relax(1, 2)
    assert np.abs(p).max() > 0
    assert _blocks_equal(p, o, "p", 4, 4) == []
    o.close()


@pytest.mark.parametrize("dims", [(32, 64, 64), (16, 32, 128), (64, 64, 32)])
def test_in_kernel_coefficients_match_stored_slots_on_stretched_grids(mg, dims, monkeypatch):
    """The level-1 smoother rebuilds the pivots, the diagonal, slots 3,5,6,8 (slopes), slots 4 / 7 and the interface depths zw
    (sigma-coordinate formula) in registers.  With a stretched coordinate (theta_s, theta_b > 0: cosh / exp tables), a moving free
    surface (zeta /= 0), non-uniform dx, dy and a non-zero hc, the result must equal, bit for bit, the same solve through the STORED
    coefficients (MGX_NO_MF=1: no in-kernel reconstruction) -- device against device, so libm differences do not enter."""
    from mgroms_amd.testcases import seamount_geometry, resting_column_state
    nx, ny, nz = dims
    dx, dy, zeta, h = seamount_geometry(nx, ny)
    ii, jj = np.meshgrid(np.arange(nx + 2), np.arange(ny + 2), indexing="ij")
    dx = dx * (1.0 + 0.2 * np.sin(0.3 * ii)); dy = dy * (1.0 + 0.1 * np.cos(0.2 * jj))
    zeta = 0.4 * np.cos(0.25 * ii) * np.sin(0.15 * jj)
    u, v, w = resting_column_state(nx, ny, nz)
    out = []
    for nomf in (False, True):
        if nomf:
            monkeypatch.setenv("MGX_NO_MF", "1")
        else:
            monkeypatch.delenv("MGX_NO_MF", raising=False)
        mg.nhydro_init(nx, ny, nz, 1, 1, 0, mg.nhydro.default_params(relax_method="FC"))
        mg.nhydro_matrices(dx, dy, zeta, h, None, 250.0, 0.4, 6.0)
        mg.nhydro.compute_rhs(u, v, w)
        n, hist = mg.solve_p(1e-12, 3)
        out.append((hist.copy(), mg.grid(1).p))
    monkeypatch.delenv("MGX_NO_MF", raising=False)
    # the fields bit for bit; the norms to 1e-13: with the stored coefficients the closing residual of an iteration is the plain kernel, with the
    # in-kernel ones it is fused with the next restriction (mgx_resrest.hip) and sums its partials in another order
    assert np.array_equal(out[0][1], out[1][1]) and np.all(np.abs(out[0][0] - out[1][0]) <= 1e-13 * out[1][0])
    assert np.abs(out[0][1]).max() > 0 and out[0][0][3] < out[0][0][0]


# ---- BASELINE config 4: rndtopo 1024x1024x64 -----------------------------------------------------------------------------
def test_config4_rndtopo_1024x1024x64_bitwise(mg):
    """mg_testrndtopo's geometry (h = 0.2*Htot*U per cell, mg_setup_tests.f90:199; seeded generator of this build) at the
    configuration's full size on one GPU against the oracle on 4x4 emulated ranks, two solve_p iterations.  The F-cycle
    DIVERGES on this input (10 m cells, depth jumps of hundreds of metres) in the oracle and on the GPU alike: a parity
    and throughput case, not a convergence case."""
    _gpu(mg, 1024, 1024, 64, geom="rndtopo", relax_method="FC")
    n, hist = mg.solve_p(1e-12, 2)
    p = mg.grid(1).p
    mg.nhydro_clean()
    o = _oracle(1024, 1024, 64, 4, 4, geom="rndtopo", relax_method="FC")
    no, ho, _ = o.solve_p(1e-12, 2)
    assert n == no == 2 and _hist_close(hist, ho), (hist, ho)
    assert _blocks_equal(p, o, "p", 4, 4) == []
    o.close()


# ---- the mask of the call (nhydro.f90:56,72): honoured with bmask off too ------------------------------------------------
@pytest.mark.parametrize("bmask,dims", [(0, (32, 32, 8)), (1, (32, 32, 8)), (0, (48, 32, 8)), (1, (16, 48, 8))])
def test_call_mask_is_used(mg, bmask, dims):
    """compute_rhs multiplies the w cross terms by the rmask handed to nhydro_solve whatever bmask says, and builds umask /
    vmask from it when bmask (mg_compute_rhs.f90:56-72,110-111): a mask that differs from the one of nhydro_matrices must
    show up in b, u, v, w exactly as in the oracle.
    nx /= ny pins the layout of the per-call mask: (0:ny+1, 0:nx+1) with j fastest, the memory the reference's drivers allocate
    and fill (mg_testseamount.f90:97).  nhydro_solve itself declares the dummy (0:nx+1,0:ny+1) and then indexes rmask(j,i)
    (nhydro.f90:56,72; mg_compute_rhs.f90:61): on a non-square block with a non-trivial mask the reference reads element
    j+(nx+2)*i of an array filled at j+(ny+2)*i.  The library follows the drivers' layout, i.e. the intent."""
    from oracle.mgoracle import Oracle, seamount_geometry
    from mgroms_amd.testcases import island_mask
    nx, ny, nz = dims
    kw = dict(relax_method="FC", solver_prec=1e-9, solver_maxiter=4)
    mg.nhydro_init(nx, ny, nz, 1, 1, 0, mg.nhydro.default_params(bmask=bmask, **kw))
    dx, dy, zeta, h = seamount_geometry(nx, ny, 1, 1, 0)
    m0 = island_mask(nx, ny)
    mg.nhydro_matrices(dx, dy, zeta, h, m0 if bmask else None, 4e3, 0.0, 0.0)
    o = Oracle(nx, ny, nz, 1, 1, bmask=bool(bmask), **kw)
    for name, a in (("dx", dx), ("dy", dy), ("zeta", zeta), ("h", h)):
        o.field(name)[...] = a
    if bmask:
        o.field("rmask")[...] = m0
    o.matrices(4e3, 0.0, 0.0)
    rng = np.random.default_rng(3)
    mcall = m0.copy()
    mcall[5:9, ny - 12:ny - 6] = 0.0  # an extra piece of land the matrices have not seen (not symmetric under i <-> j)
    u = rng.uniform(-1, 1, (nz, ny + 2, nx + 1)); v = rng.uniform(-1, 1, (nz, ny + 1, nx + 2)); w = rng.uniform(-1, 1, (nz + 1, ny + 2, nx + 2))
    o.field("u")[...] = u; o.field("v")[...] = v; o.field("w")[...] = w
    o.field("rmaska")[...] = mcall
    o.use_call_mask(True)
    mg.nhydro.compute_rhs(u, v, w, mcall)
    o.compute_rhs()
    b = mg.grid(1).b
    assert np.array_equal(b, o.field("b"))
    mg.nhydro.compute_rhs(u, v, w, None)  # without a per-call mask the result is a different one
    assert not np.array_equal(mg.grid(1).b, b)
    mg.nhydro_solve(u, v, w, mcall)
    o.nhydro_solve()
    assert np.array_equal(mg.grid(1).p, o.field("p"))
    assert np.array_equal(u, o.field("u")) and np.array_equal(v, o.field("v")) and np.array_equal(w, o.field("w"))


def test_native_rccl_world_size_one(mg):
    """libmgx.so's own RCCL communicator (include/mgx.h: mgx_rccl_*) on a world of one rank: bootstrap, the rank-coded
    exchange and the all-reduce of mgx_rccl_selftest -- the same hooks a multi-GPU halo fill uses.  (RCCL refuses several
    ranks on one device, so this is what a one-GPU box can run.)"""
    import ctypes as C
    from mgroms_amd._lib import lib, check
    L = lib()
    mg.nhydro_clean()
    blob = C.create_string_buffer(L.mgx_rccl_unique_id_bytes())
    check(L.mgx_rccl_get_unique_id(blob))
    check(L.mgx_rccl_connect(blob, 1, 0))
    try:
        _gpu(mg, 32, 32, 8, relax_method="FC")
        assert b"RCCL, native" in L.mgx_transport()
        check(L.mgx_rccl_selftest())
        n, hist = mg.solve_p(1e-8, 20)
        assert hist[-1] <= 1e-8
    finally:
        L.mgx_rccl_disconnect()
        mg.nhydro_clean()
