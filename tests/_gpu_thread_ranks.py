"""All ranks of a multi-rank GPU solve as THREADS of this one process (tests/test_gpu_multirank.py): each thread selects its own
libmgx.so instance (include/mgx.h: mgx_instance_*), its own HIP stream on cuda:0 and an mgroms_amd.parallel.ThreadComm.  This
is how BASELINE config 5's 4x2 process grid (8 ranks, a 2x2 gather followed by a 2x1 gather, mg_grids.f90:543-565,702-718) runs on
the one GPU of a test box, which admits at most 6 processes on its card.  Every rank's block is compared bit for bit with the
oracle's emulated MPI ranks, through the peer-to-peer pushes and again through the mgx_set_comm hooks.

usage: _gpu_thread_ranks.py npx npy nx ny nz nsmall [drop]     (drop: one rank goes silent for one exchange first, see rank_main)"""
import os
import sys
import threading
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_main(rank, tw, cfg, o, ho, no, results):
    import torch
    import mgroms_amd as mg
    from mgroms_amd import nhydro
    from mgroms_amd._lib import check, lib
    from mgroms_amd.parallel import ThreadComm
    from oracle.mgoracle import seamount_geometry
    npx, npy, nx, ny, nz, nsmall, tol, maxit, nsc, drop = cfg

    def stage(what):
        print(f"[rank {rank}] {what}", file=sys.stderr, flush=True)
    try:
        torch.cuda.set_device(0)
        torch.cuda.set_stream(torch.cuda.Stream())
        L = lib()
        inst = L.mgx_instance_create()
        check(L.mgx_instance_select(inst))
        assert L.mgx_instance_current() == inst
        nhydro.set_verbose(0)
        comm = ThreadComm(tw, rank, p2p=True)
        par = nhydro.default_params(relax_method="FC", solver_prec=tol, nsmall=nsmall, ns_coarsest=nsc)
        mg.nhydro_init(nx, ny, nz, npx, npy, rank, par, comm=comm)
        stage("init done, p2p_active=%s" % comm.p2p_active)
        assert comm.p2p_active, comm.p2p_error
        dx, dy, zeta, h = seamount_geometry(nx, ny, npx, npy, rank)
        mg.nhydro_matrices(dx, dy, zeta, h, None, 4e3, 0.0, 0.0)
        u = np.zeros((nz, ny + 2, nx + 1)); v = np.zeros((nz, ny + 1, nx + 2)); w = -np.ones((nz + 1, ny + 2, nx + 2)); w[0] = 0
        nhydro.compute_rhs(u, v, w)
        stage("matrices + rhs done")
        if drop:
            # A rank goes silent for one exchange (test hook: its flags stay down): its neighbours' waits time out (200 ms here).  Nobody
            # may fall back alone -- the ranks agree at the next norm and ALL of them get the same error and the same transport.
            from mgroms_amd._lib import MgxError
            nhydro.set_option("p2p_timeout_ms", 200)
            tw.barrier.wait(60)
            if rank == tw.world - 1:
                nhydro.set_option("p2p_test_drop", 7)
            try:
                mg.solve_p(tol, maxit)
                raise AssertionError("the solve with a silent rank did not fail")
            except MgxError as e:
                assert "ALL ranks have switched to the hooks together" in str(e), str(e)
            stage("collective fallback seen")
            assert "peer-to-peer" not in comm.transport(), comm.transport()
            before = nhydro.counters()["p2p_exchanges"]
            n, hist = mg.solve_p(tol, maxit)          # the repeat, through the hooks on every rank
            assert nhydro.counters()["p2p_exchanges"] == before
            assert n == no and np.array_equal(mg.grid(1).p, o.field("p", 1, rank)), rank
            comm.set_p2p(True)                         # and the pushes can be switched on again, all ranks together
            nhydro.set_option("p2p_timeout_ms", 5000)
        n, hist = mg.solve_p(tol, maxit)
        stage("solve done")
        assert mg.nlevs() == o.nlevs
        gathered = [l for l in range(1, o.nlevs + 1) if o.level_info(l, rank)["gather"]]
        for lev in range(1, o.nlevs + 1):
            g = mg.grid(lev)
            li = o.level_info(lev, rank)
            assert (g.nx, g.ny, g.nz, g.npx, g.npy, g.gather) == (li["nx"], li["ny"], li["nz"], li["npx"], li["npy"], li["gather"]), (rank, lev)
            for name in ("h", "zr", "cA"):
                assert np.array_equal(g.get(name), o.field(name, lev, rank)), (rank, lev, name)
        assert np.array_equal(mg.grid(1).b, o.field("b", 1, rank)), rank
        assert n == no, (n, no)
        p_first = mg.grid(1).p
        assert np.array_equal(p_first, o.field("p", 1, rank)), rank
        # same reduction values on every rank, summed in another order than the oracle's: 1e-12 (ADVICE: the oracle comparison stays tight)
        assert np.all(np.abs(hist - ho) <= 1e-14 + 1e-12 * np.abs(ho)), (hist, ho)
        c = nhydro.counters()
        assert c["p2p_exchanges"] > 0 and c["exchanges"] > 0
        # every level's p after the solve, gathered levels included (their blocks are the same on every member of a group)
        for lev in range(2, o.nlevs + 1):
            assert np.array_equal(mg.grid(lev).p, o.field("p", lev, rank)), (rank, lev)
        # the same solve through the hooks instead of the pushes: identical iterates
        stage("fields compared")
        comm.set_p2p(False)
        n2, hist2 = mg.solve_p(tol, maxit)
        stage("second solve (hooks) done")
        assert n2 == n and np.array_equal(hist2, hist) and np.array_equal(mg.grid(1).p, p_first), rank
        assert nhydro.counters()["p2p_exchanges"] == c["p2p_exchanges"]
        comm.set_p2p(True)
        # the prolongation's first-colour shortcut on OPEN-sided levels (neighbours instead of mirrors): same bits without it
        nhydro.set_option("c2f_skip", 0)
        n3, hist3 = mg.solve_p(tol, maxit)
        nhydro.set_option("c2f_skip", 1)
        assert n3 == n and np.array_equal(hist3, hist) and np.array_equal(mg.grid(1).p, p_first), rank
        # mg_testhalo.f90:75-92 on the 8-neighbour topology: the halo planes hold the neighbours' rank numbers
        g1 = mg.grid(1)
        g1.set("p", np.full(g1._shape("p"), float(rank)))
        mg.fill_halo(1, "p")
        ph = g1.get("p")
        nbr = g1.neighb
        want = lambda q: float(q if q >= 0 else rank)
        assert np.all(ph[1:-1, 0, :] == want(nbr[0])) and np.all(ph[-1, 1:-1, :] == want(nbr[1])), rank
        assert np.all(ph[1:-1, -1, :] == want(nbr[2])) and np.all(ph[0, 1:-1, :] == want(nbr[3])), rank
        for c4, d in (((0, 0), 4), ((-1, 0), 5), ((-1, -1), 6), ((0, -1), 7)):
            if nbr[d] >= 0:
                assert np.all(ph[c4[0], c4[1], :] == float(nbr[d])), (rank, d)
        tw.barrier.wait(120)
        mg.nhydro_clean()
        check(L.mgx_instance_select(0))
        check(L.mgx_instance_destroy(inst))
        results[rank] = f"rank {rank} ok nite={n} gathered_levels={gathered} p2p_exchanges={c['p2p_exchanges']}"
    except BaseException:
        results[rank] = "rank %d FAILED:\n%s" % (rank, traceback.format_exc())
        try:
            tw.barrier.abort()
        except Exception:
            pass


def main():
    npx, npy, nx, ny, nz, nsmall = (int(a) for a in sys.argv[1:7])
    world = npx * npy
    os.environ["OMP_NUM_THREADS"] = "8"
    # One hardware queue per rank.  The HIP runtime multiplexes the streams of a process onto GPU_MAX_HW_QUEUES hardware queues
    # (default 4): two ranks whose streams share a queue are serialised, and a halo kernel that waits for its neighbour's push would sit
    # in front of the very kernel that pushes.  (Process-per-rank runs have a queue set each.)  Must be set before HIP initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, 3 * world)))  # each rank drives two streams (the exchange runs beside the interior sweep)
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("MGX_TEST_WATCHDOG", "100")), exit=True)  # a stuck collective: all stacks, then exit
    import torch
    torch.cuda.set_device(0)
    from mgroms_amd.parallel import ThreadWorld
    from oracle.mgoracle import make_seamount
    tol, maxit, nsc = 1e-9, 3, 6
    o = make_seamount(nx, ny, nz, npx, npy, relax_method="FC", solver_prec=tol, nsmall=nsmall, ns_coarsest=nsc)
    o.compute_rhs()
    no, ho, _ = o.solve_p(tol, maxit)
    tw = ThreadWorld(world)
    results = [None] * world
    cfg = (npx, npy, nx, ny, nz, nsmall, tol, maxit, nsc, len(sys.argv) > 7 and sys.argv[7] == "drop")
    th = [threading.Thread(target=rank_main, args=(r, tw, cfg, o, ho, no, results), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(90)
    bad = [r for r in range(world) if results[r] is None or "ok" not in results[r].split("\n")[0]]
    for r in range(world):
        print(results[r] if results[r] is not None else f"rank {r} did not finish")
    sys.stdout.flush()
    os._exit(1 if bad else 0)  # daemon threads may still sit in a collective after a failure


if __name__ == "__main__":
    main()
