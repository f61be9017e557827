#!/usr/bin/env python3
"""Headline benchmark: multigrid V-cycles per second of the pressure solve on the seamount 512x512x64 problem
(BASELINE.json `metric`, configs[2]: the configuration the metric is quoted on; it fits one GPU), plus the
HBM roofline fraction of the level-1 smoother kernel and the CPU baseline (oracle) timed on the host cores.

  python bench.py --gpus N --steps K --warmup W        (N>1: one rank per GPU; under torch.distributed.run the ranks are
                                                        taken from the environment, from a bare shell bench.py starts them itself)

A step = one Vcycle(1) (mg_solvers.f90:129) over the resident fields: ns_pre=3 sweeps + residual + restriction on
every level down, 40 sweeps on the coarsest, prolongation + ns_post=2 sweeps up.  Inputs are synthetic (the
reference's seamount geometry, u=v=0, w=-1) and resident in HBM before the timed region.  Weak scaling: every rank
owns a 512x512x64 block; `value` is V-cycles/s scaled by (global cells / cells of one block), i.e. block-V-cycles/s.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
SMOOTHER_BYTES_PER_CELL = 88  # cA 64 + b 8 + p read 8 + p write 8, per cell per full sweep
VCYCLE_BYTES_PER_CELL = 633   # one Vcycle(1), all levels
FCYCLE_BYTES_PER_CELL = 812   # one solve_p iteration = Fcycle + the closing residual
# the reference's own value of `nsmall` for the named multi-GPU configurations: the namelist default 8 (config 4), 16 for config 5
REFERENCE_NSMALL = {2: 8, 4: 8, 8: 16}
PGRID = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}
# oracle/mgoracle.c against the real reference (flang -O2 + MPICH) on the SAME 8 cores of the development container, 4x2
# ranks, seamount 512x512x64: reference time / port time.  Relates the port's number on the
# GPU box's host cores back to the reference, which cannot travel.  The port's side of each ratio is re-measured by
# scripts/cpu_calibration.py (run in the development container); the reference's side is BASELINE.md's recorded figure.
CALIBRATION = {"host": "development container, 8 cores, 4x2 (emulated) ranks, seamount 512x512x64",
               "FC": {"solve_iteration": 0.66 / 0.711, "level1_sweep": 75.0 / 92.8, "level1_residual": 36.0 / 48.3},
               "RB": {"solve_iteration": 0.68 / 0.750, "level1_sweep": 67.0 / 75.9, "level1_residual": 44.0 / 42.7}}


def cpu_baseline(nx, ny, nz, method, cycles=2):
    """The oracle (CPU restatement of the reference, oracle/mgoracle.c) on the same workload, decomposed over the
    host cores like the reference's MPI ranks (one OpenMP thread per emulated rank)."""
    from oracle.mgoracle import make_seamount
    ncpu = os.cpu_count() or 1
    cores = 1
    while cores * 2 <= min(ncpu, 16):
        cores *= 2
    npx, npy = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2), 16: (4, 4)}[cores]
    os.environ["OMP_NUM_THREADS"] = str(cores)
    o = make_seamount(nx // npx, ny // npy, nz, npx, npy, relax_method=method)
    o.compute_rhs()
    o.vcycle(1)  # untimed: first touch
    t0 = time.perf_counter()
    for _ in range(cycles):
        o.vcycle(1)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    o.relax(1, 1)
    sweep = time.perf_counter() - t1
    o.close()
    return {"value": cycles / dt, "unit": "Vcycle(1)/s", "cores": cores, "kind": "port",
            "calibration_vs_reference": CALIBRATION[method]["solve_iteration"],
            "calibration": {"what": "reference time / port time on the same cores (1.0 = as fast as the reference)",
                            "host": CALIBRATION["host"], **CALIBRATION[method]},
            "sample": f"{cycles} Vcycle(1) of seamount {nx}x{ny}x{nz} {method} on {npx}x{npy} emulated ranks "
                      f"({cores} OpenMP threads), after 1 untimed cycle",
            "level1_sweep_ms": sweep * 1e3,
            "level1_sweep_GBs": SMOOTHER_BYTES_PER_CELL * nx * ny * nz / sweep / 1e9}


def live_traffic(nx, ny, nz, method, kname):
    """HBM bytes per launch of the level-1 colour pass from the PMC counters, measured in THIS invocation: two rocprofv3 passes (FETCH_SIZE,
    WRITE_SIZE -- they do not fit one pass) over a child process that runs a few level-1 sweeps of the same workload (scripts/sweep_time.py),
    started BEFORE this process touches the GPU; FETCH x 2 (the guide's gfx950 correction, scripts/pmc_summary.py), WRITE exact, the largest
    dispatch of the kernel.  None when rocprofv3 is missing, this process is itself being profiled, or a pass fails / times out: the line then
    falls back to the committed capture and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None or any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    vals = {}
    tmp = tempfile.mkdtemp(prefix="mgx_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.join(ROOT, "scripts", "sweep_time.py"), str(nx), str(ny), str(nz), method, "3"]
            # (a process group of its own: a pass that times out is ended together with the profiled child, so nothing keeps the GPU busy behind the bench)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=240)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                return None
            files = glob.glob(os.path.join(out, "*", "*_counter_collection.csv"))
            if rc != 0 or not files:
                return None
            best = 0.0
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == counter and row["Kernel_Name"].split("(")[0].replace("void ", "") == kname:
                    best = max(best, float(row["Counter_Value"]))
            if best <= 0.0:
                return None
            vals[counter] = best
        return 2.0 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _spawn_ranks(n):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as fresh child processes through
    torch.distributed.run BEFORE this process touches the GPU (it never does), pass their output through and exit with
    their status.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, nargs=3, default=[512, 512, 64], help="local block nx ny nz")
    ap.add_argument("--method", default="FC", choices=["FC", "RB"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep-reps", type=int, default=20)
    ap.add_argument("--nsmall", type=int, default=256,
                    help="nsmall of the timed region when N>1 (the reference's coarse-level agglomeration threshold, a namelist member; its default is 8). "
                         "The rate at the reference's own value for the named configuration is measured as well (also_reference_nsmall)")
    ap.add_argument("--no-p2p", action="store_true", help="N>1: halos through the torch.distributed callback only")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 path with all ranks on cuda:0 of a one-GPU box (host-staged transport)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this invocation (two rocprofv3 --pmc passes over a child process, ~20 s): read the committed capture")
    ap.add_argument("--native-rccl", action="store_true",
                    help="N>1: libmgx.so's own RCCL communicator instead of the torch.distributed callbacks (opt-in: never run on more than one GPU yet)")
    args = ap.parse_args()
    if args.gpus not in PGRID:
        raise SystemExit("--gpus must be 1, 2, 4 or 8")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _spawn_ranks(args.gpus)

    # N = 1: the dominant kernel's HBM traffic from the PMC counters, measured now, in child processes, before this process touches the GPU
    live_kname = (f"k_relax_nz<{args.size[2]}, true, {'true' if args.method == 'RB' else 'false'}, 3, true, true>" if args.size[2] == 64 else None)
    traffic_live = None
    if args.gpus == 1 and not args.no_live_traffic and live_kname and "WORLD_SIZE" not in os.environ:
        traffic_live = live_traffic(*args.size, args.method, live_kname)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        from mgroms_amd.parallel import Comm
        comm = Comm(p2p=not args.no_p2p, native=(args.backend == "nccl" and args.native_rccl))

    import mgroms_amd as mg
    from mgroms_amd import nhydro
    from mgroms_amd.testcases import seamount_geometry, resting_column_state

    nx, ny, nz = args.size
    npx, npy = PGRID[args.gpus]
    nhydro.set_verbose(0)
    # N>1: agglomerate the launch-/latency-bound coarse levels (local size < 256) with the reference's own knob
    # `nsmall` (mg_namelist.f90:11, mg_grids.f90:550): levels 3..6 are then computed redundantly on fewer ranks and
    # need few or no halo exchanges (each exchange costs more than a whole colour pass there).  FC results do not
    # depend on the decomposition (tests/test_oracle.py::test_fc_is_decomposition_independent_on_4x2), so this is
    # the same solve as nsmall=8.
    par = nhydro.default_params(relax_method=args.method, nsmall=(8 if world == 1 else args.nsmall))
    mg.nhydro_init(nx, ny, nz, npx, npy, rank, par, comm=comm)
    dx, dy, zeta, h = seamount_geometry(nx, ny, npx, npy, rank)
    mg.nhydro_matrices(dx, dy, zeta, h, None, 4e3, 0.0, 0.0)
    u, v, w = resting_column_state(nx, ny, nz)
    nhydro.compute_rhs(u, v, w)
    res0 = mg.compute_residual(1)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # the K timed steps are enqueued back to back (option "async": Vcycle returns without waiting for the stream, as a GPU-resident model
    # calls it) and the region closes with the barrier + synchronize of the contract, after which the library reports device-side errors
    nhydro.set_option("async", 1)
    for _ in range(args.warmup):
        mg.Vcycle(1)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mg.Vcycle(1)
    sync()
    dt = time.perf_counter() - t0
    nhydro.synchronize()
    nhydro.set_option("async", 0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res1 = mg.compute_residual(1)
    # the same K steps with a host synchronisation after every cycle (how rounds 1 and 2 timed the step): reported beside `value`
    sync()
    t0s = time.perf_counter()
    for _ in range(args.steps):
        mg.Vcycle(1)
    sync()
    dt_sync_each = time.perf_counter() - t0s
    if world > 1:
        t = torch.tensor([dt_sync_each], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_sync_each = float(t.item())

    # N>1: the peer-to-peer halo transport is only trusted if, on THIS machine, it reproduces the torch.distributed
    # (RCCL) transport bit for bit (four-colour iterates do not depend on who moves the halos); otherwise the timed
    # region is repeated through the RCCL callback and that number is reported.
    transport, transport_check = None, None
    if world > 1:
        transport = comm.transport()
        base_transport = ("RCCL, native in libmgx.so" if comm.native_active else f"torch.distributed callbacks ({args.backend})")
        if not comm.p2p_active:
            transport_check = f"p2p unavailable: {comm.p2p_error}"
        else:
            def two_cycles():
                mg.solve_p(1e30, 0)  # p = 0 (cold start, mg_solvers.f90:35), no iteration
                mg.Vcycle(1)
                mg.Vcycle(1)
                return mg.compute_residual(1)
            r_p2p = two_cycles()
            comm.set_p2p(False)
            r_cb = two_cycles()
            same = torch.tensor([1 if r_p2p == r_cb else 0], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            if int(same.item()) == 1:
                transport_check = f"residual after 2 V-cycles identical through the pushes and through {base_transport} ({r_p2p:.17g})"
                comm.set_p2p(True)
            else:
                transport_check = f"p2p REJECTED: residual {r_p2p:.17g} vs {r_cb:.17g} through {base_transport}; timing repeated through the latter"
                transport = comm.transport()
                for _ in range(args.warmup):
                    mg.Vcycle(1)
                sync()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    mg.Vcycle(1)
                sync()
                dt = time.perf_counter() - t0
                t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())

    # roofline of the dominant kernel: the level-1 smoother sweep (HIP events on the solver's stream)
    sweep_ms = nhydro.time_relax(1, args.sweep_reps)
    resid_ms = nhydro.time_residual(1, args.sweep_reps)
    ncol = 4 if args.method == "FC" else 2
    cells = nx * ny * nz
    launch_bytes = SMOOTHER_BYTES_PER_CELL * cells / ncol
    achieved = launch_bytes / (sweep_ms / ncol * 1e-3) / 1e9
    # F-cycle iteration rate (solve_p iteration = Fcycle + residual), for reference.  Fcycle is enqueued (option "async"), the residual
    # norm that follows synchronises -- the one host synchronisation per iteration solve_p itself has
    sync()
    nhydro.set_option("async", 1)
    t2 = time.perf_counter()
    nf = max(5, args.steps // 2)
    for _ in range(nf):
        mg.Fcycle()
        mg.compute_residual(1)
    sync()
    fc_rate = nf / (time.perf_counter() - t2)
    nhydro.synchronize()
    nhydro.set_option("async", 0)
    # the same through solve_p itself (mg_solvers.f90:17-101: one host synchronisation per iteration, for the norm), p restored afterwards
    p_keep_sp = mg.grid(1).p
    nhydro.set_option("warm_start", 1)
    sync()
    t2b = time.perf_counter()
    nsp = max(10, args.steps // 2)  # (the call's own set-up -- ||b|| and the first residual -- is 1 % of ten iterations)
    mg.solve_p(1e-300, nsp)
    sync()
    sp_rate = nsp / (time.perf_counter() - t2b)
    nhydro.set_option("warm_start", 0)
    mg.grid(1).set("p", p_keep_sp)

    # The same two rates with the coarsest-level solve of a cycle as ONE matrix-vector product (option "coarsest_direct" = 2): the same
    # linear map as the ns_coarsest sweeps in another association -- within 1e-12, NOT bit-identical -- so it is not what `value` measures (four
    # colours keep their bit parity by default); reported beside it.  N = 1 only (a closed, un-gathered coarsest level).
    also_direct = None
    if world == 1 and int(os.environ.get("MGX_COARSEST_DIRECT", "1")) != 2:
        nhydro.set_option("coarsest_direct", 2)
        nhydro.set_option("async", 1)
        p_keep_cd = mg.grid(1).p
        for _ in range(2):
            mg.Vcycle(1)
        sync()
        tcd = time.perf_counter()
        for _ in range(args.steps):
            mg.Vcycle(1)
        sync()
        cd_ms = (time.perf_counter() - tcd) / args.steps * 1e3
        tcd = time.perf_counter()
        for _ in range(nf):
            mg.Fcycle()
            mg.compute_residual(1)
        sync()
        cd_fc_rate = nf / (time.perf_counter() - tcd)
        nhydro.synchronize()
        nhydro.set_option("async", 0)
        also_direct = {"vcycles_per_sec": 1e3 / cd_ms, "ms_per_step": cd_ms, "fcycle_iterations_per_sec": cd_fc_rate,
                       "vcycle_roofline_frac": VCYCLE_BYTES_PER_CELL * cells / (cd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "fcycle_roofline_frac": FCYCLE_BYTES_PER_CELL * cells * cd_fc_rate / 1e9 / HBM_PEAK_GBS,
                       "direct_solves": nhydro.get_option("coarsest_direct_solves"),
                       "tolerance": "every level's p within 1e-12 of the sweeps' (tests/test_gpu_parity.py::test_coarsest_solve_as_one_matrix_vector_product); "
                                    "the default (1) uses it for red-black in the sequential order only, so `value` is the bit-exact four-colour cycle"}
        nhydro.set_option("coarsest_direct", 1)
        mg.grid(1).set("p", p_keep_cd)
    counters_main, nlev_main = nhydro.counters(), mg.nlevs()
    # N>1: what one V-cycle exchanges on every rank, and what a level-1 / level-2 halo fill costs there (enqueued back to back, HIP stream time)
    exchange_report = None
    if world > 1:
        c0 = nhydro.counters()
        mg.Vcycle(1)
        c1 = nhydro.counters()
        per_cycle = {k: c1[k] - c0[k] for k in ("halo_fills", "exchanges", "p2p_exchanges", "launches") if k in c0}
        fill_us = {}
        nhydro.set_option("async", 1)
        for lev in (1, 2):
            if lev > mg.nlevs():
                continue
            for _ in range(5):
                mg.fill_halo(lev, "p")
            sync()
            tq = time.perf_counter()
            for _ in range(50):
                mg.fill_halo(lev, "p")
            sync()
            fill_us[f"level{lev}"] = (time.perf_counter() - tq) / 50 * 1e6
        nhydro.synchronize()
        nhydro.set_option("async", 0)
        mine = {"rank": rank, "per_vcycle": per_cycle, "halo_fill_us": fill_us}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        exchange_report = allr
    # N>1: the same step at the reference's own nsmall for this configuration (every coarse level distributed; config 5: the 2x2 and 2x1 gathers)
    also_ref_nsmall = None
    if world > 1 and args.nsmall != REFERENCE_NSMALL[world]:
        mg.nhydro_init(nx, ny, nz, npx, npy, rank, nhydro.default_params(relax_method=args.method, nsmall=REFERENCE_NSMALL[world]), comm=comm)
        mg.nhydro_matrices(dx, dy, zeta, h, None, 4e3, 0.0, 0.0)
        nhydro.compute_rhs(u, v, w)
        nhydro.set_option("async", 1)
        for _ in range(2):
            mg.Vcycle(1)
        sync()
        tr = time.perf_counter()
        nr = max(5, args.steps // 2)
        for _ in range(nr):
            mg.Vcycle(1)
        sync()
        dtr = time.perf_counter() - tr
        nhydro.synchronize()
        nhydro.set_option("async", 0)
        t = torch.tensor([dtr], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dtr = float(t.item())
        also_ref_nsmall = {"nsmall": REFERENCE_NSMALL[world], "vcycles_per_sec": nr / dtr * npx * npy, "ms_per_step": dtr / nr * 1e3, "levels": mg.nlevs(),
                           "halo_transport": comm.transport()}
    # the reference's default ordering (red-black, cmatrix='real') on the same workload (N=1 only), in its three modes: the sequential order at
    # speed (the default: parallel pass + scan over the planes + rank-one correction, within 1e-10 of the reference's loop), the plain
    # parallel sweep (5e-5 away) and -- one sweep, once -- the bit-exact order with a launch per plane
    also_rb = None
    if world == 1 and args.method == "FC":
        also_rb = {"note": "relax_method='RB', cmatrix='real' (the reference default, BASELINE config 2's ordering) on the bench's workload"}
        for mode, opts, tol in (("sequential_order", {"rb_seq": 1, "rb_exact": 0}, "history and p within 1e-10 of the reference's sequential loop (a few ulp per sweep; tests: 1e-12 per relax call)"),
                                ("parallel", {"rb_seq": 0, "rb_exact": 0}, "history within 5e-5 relative, same iteration counts (old same-colour k=1 diagonals everywhere)")):
            for k, val in opts.items():
                nhydro.set_option(k, val)
            mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method="RB"))
            mg.nhydro_matrices(dx, dy, zeta, h, None, 4e3, 0.0, 0.0)
            nhydro.compute_rhs(u, v, w)
            nhydro.set_option("async", 1)
            for _ in range(2):
                mg.Vcycle(1)
            sync()
            t3 = time.perf_counter()
            nrb = max(5, args.steps // 2)
            for _ in range(nrb):
                mg.Vcycle(1)
            sync()
            rb_ms = (time.perf_counter() - t3) / nrb * 1e3
            nhydro.synchronize()
            nhydro.set_option("async", 0)
            rb_sweep = nhydro.time_relax(1, args.sweep_reps)
            sync()
            t4 = time.perf_counter()
            nit_rb = 5
            nhydro.set_option("warm_start", 1)
            mg.solve_p(1e-300, nit_rb)
            sync()
            rb_it_ms = (time.perf_counter() - t4) / nit_rb * 1e3
            nhydro.set_option("warm_start", 0)
            if mode == "sequential_order":
                window_info = {f"level{l}": dict(zip(("rho", "planes"), nhydro.rbseq_window_info(l))) for l in range(1, mg.nlevs() + 1)}
                window_info["colours_done_by_the_windowed_walk"] = nhydro.get_option("rbseq_window_colours")
                for l in range(1, mg.nlevs() + 1):
                    window_info[f"level{l}"]["rows_corrected"] = nhydro.rbseq_window_rows(l)
            also_rb[mode] = {"vcycles_per_sec": 1e3 / rb_ms, "ms_per_step": rb_ms, "sweep_ms": rb_sweep, "solve_p_iteration_ms": rb_it_ms,
                             "roofline_frac": SMOOTHER_BYTES_PER_CELL * nx * ny * nz / (rb_sweep * 1e-3) / 1e9 / HBM_PEAK_GBS, "tolerance": tol}
        nhydro.set_option("rb_exact", 1)
        mg.nhydro_init(nx, ny, nz, 1, 1, 0, nhydro.default_params(relax_method="RB"))
        mg.nhydro_matrices(dx, dy, zeta, h, None, 4e3, 0.0, 0.0)
        nhydro.compute_rhs(u, v, w)
        nhydro.time_relax(1, 1)
        also_rb["exact_order"] = {"sweep_ms": nhydro.time_relax(1, 1), "tolerance": "bit-identical to the reference's sequential loop (one launch per plane: a parity mode)"}
        nhydro.set_option("rb_exact", 0); nhydro.set_option("rb_seq", 1)
        also_rb["sweep_ratio_sequential_over_parallel"] = also_rb["sequential_order"]["sweep_ms"] / also_rb["parallel"]["sweep_ms"]
        also_rb["vcycle_ratio_sequential_over_parallel"] = also_rb["sequential_order"]["ms_per_step"] / also_rb["parallel"]["ms_per_step"]
        also_rb["sequential_order"]["how"] = ("per colour: parallel pass (writes the walk's d0), then ONE launch in which every workgroup walks the m planes in front of its own from zero "
                                              "(the walk contracts by rho per plane, rho^m <= 2^-64: option rbseq_window) and corrects its columns up to the row where T^-1 e1 has decayed to 2^-64 "
                                              "(option rbseq_rowcut); the coarsest solve of a cycle as one matrix-vector product (coarsest_direct); DESIGN.md 4.4, 4.6")
        also_rb["sequential_order"]["window"] = window_info

    # HBM traffic of the dominant kernel from the PMC counters: they cannot be collected inside this run (rocprofv3 must wrap
    # the process, in passes of their own), so the figure is read from the committed capture of THIS command
    # (profiles/r02_pmc_traffic.json, made by scripts/pmc_summary.py) and labelled as such
    traffic, traffic_src = None, None
    import glob
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    pmc = pmcs[-1] if pmcs else ""
    # the level-1 colour pass of the headline workload (nz = 64, level too large for the Infinity Cache); other --size values run
    # other kernels (k_relax_ks for nz <= 32 when a colour has <= 512 waves, k_relax_tall for nz = 128)
    kname = f"k_relax_nz<{nz}, true, {'true' if args.method == 'RB' else 'false'}, 3, true, true>" if nz == 64 else f"level-1 colour pass, nz={nz}"
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
            if pj.get("cells") == cells and kname in pj["kernels"]:
                traffic = pj["kernels"][kname]["hbm_bytes"]
                traffic_src = (f"NOT measured in this run: profiles/{os.path.basename(pmc)}, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                               "command (scripts/capture_profiles.sh), FETCH x2 (the guide's gfx950 correction, calibrated on an 8-B/lane read of known size)")
        except Exception:
            pass
    if traffic_live is not None:
        traffic = traffic_live
        traffic_src = ("measured in this invocation: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate, with --kernel-trace only) over a child process running "
                       "level-1 sweeps of the same workload (scripts/sweep_time.py) before the timed process touched the GPU; FETCH x2 (the guide's gfx950 correction, "
                       "calibrated in profiles/r04_pmc_traffic.json on an 8-B/lane read of known size), WRITE exact; the largest dispatch of the kernel")
    out = None
    if rank == 0:
        scale = npx * npy
        out = {
            "metric": "V-cycles/sec, seamount 512x512x64 (fine-grid smoother HBM GB/s and % of 8 TB/s: see roofline)",
            "value": args.steps / dt * scale, "unit": "V-cycles/s (x number of 512x512x64 blocks when N>1: weak scaling)",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "enqueue": "async (the K steps are enqueued back to back; the region closes with barrier + synchronize)",
            "value_sync_each_step": args.steps / dt_sync_each * scale, "ms_per_step_sync_each_step": dt_sync_each / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"seamount {nx * npx}x{ny * npy}x{nz} ({npx}x{npy} ranks of {nx}x{ny}x{nz}), "
                                   f"relax_method={args.method}, ns_pre=3 ns_post=2 ns_coarsest=40, cmatrix=real, interp=linear",
                       "levels": nlev_main, "step": "one Vcycle(1)", "halo_transport": transport,
                       "native_rccl": (None if comm is None else (True if comm.native_active else f"off: {comm.native_error}")),
                       "transport_check": transport_check, "nsmall": (8 if world == 1 else args.nsmall),
                       "nsmall_note": (None if world == 1 else f"timed region: nsmall={args.nsmall} (coarse levels agglomerated early; a namelist member of the reference, "
                                       f"default 8); the reference's own value for this configuration, {REFERENCE_NSMALL[world]}, is timed in also_reference_nsmall")},
            "vcycle_roofline": {"bytes_per_cell": VCYCLE_BYTES_PER_CELL, "ms": dt / args.steps * 1e3,
                                "frac": VCYCLE_BYTES_PER_CELL * cells / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
            # what solve_p iterates: one F-cycle + the closing residual norm (mg_solvers.f90:61-68), through solve_p itself
            "fcycle_roofline": {"bytes_per_cell": FCYCLE_BYTES_PER_CELL, "ms": 1e3 / sp_rate, "frac": FCYCLE_BYTES_PER_CELL * cells * sp_rate / 1e9 / HBM_PEAK_GBS,
                                "unit": "one solve_p iteration (Fcycle + closing residual norm), SURVEY 8(d): 812 B per fine cell"},
            "exchanges": exchange_report, "also_reference_nsmall": also_ref_nsmall,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         # two labelled rates: effective = algorithmic bytes / time (= achieved, the contract's figure); measured = counter
                         # traffic / time, what the HBM interface actually carried (lower: the pass streams 7 of the 11 arrays the yardstick counts)
                         "effective_GBs": achieved,
                         "measured_GBs": (None if traffic is None else traffic / (sweep_ms / ncol * 1e-3) / 1e9),
                         "measured_frac": (None if traffic is None else traffic / (sweep_ms / ncol * 1e-3) / 1e9 / HBM_PEAK_GBS),
                         "kernel": kname + " (level-1 colour pass)",
                         "algorithmic_bytes_per_launch": launch_bytes,
                         "traffic_note": "counter traffic below the algorithmic bytes: SURVEY 8(d)'s 88 B/cell counts 11 streamed arrays; the pass "
                                         "rebuilds the pivots, slots 4/7 (from regenerated zw) and its own slopes in registers and streams 7 (DESIGN.md section 4)",
                         "launch_ms": sweep_ms / ncol, "sweep_ms": sweep_ms},
            "residual_kernel": {"ms": resid_ms, "GBs": 88 * cells / (resid_ms * 1e-3) / 1e9},
            "fcycle_iterations_per_sec": fc_rate,
            "solve_p_iterations_per_sec": sp_rate,
            "residual_before": res0, "residual_after": res1,
            "counters": counters_main,
            "also_rb": also_rb, "also_coarsest_direct": also_direct,
        }
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nx, ny, nz, args.method)
        else:
            out["cpu_baseline"] = None
    mg.nhydro_clean()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
