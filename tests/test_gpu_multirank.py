"""Multi-rank solves on the GPU: (i,j) decomposition, halo exchange (peer-to-peer pushes through hipIpc-shared device
memory by default, the torch.distributed callback in one case and in the transport cross-check of every FC case), norm
all-reduce and coarse-level gather/split, each rank checked bit for bit against the oracle's emulated MPI ranks
(four-colour ordering).  Ranks share the one
GPU of the test box (<= 4 ranks + this process)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("npx,npy,nx,ny,nz,nsmall,method", [
    (2, 1, 32, 32, 8, 8, "FC"),     # 2x1: E/W exchange only, 2x1 gather on the coarsest level
    (1, 2, 16, 32, 8, 8, "FC"),     # 1x2: N/S exchange, ragged block
    (2, 2, 32, 32, 16, 8, "FC"),    # BASELINE 64x64x16 on 2x2: 8 neighbours + corners, 2x2 gather at level 4
    (2, 2, 32, 32, 16, 32, "FC"),   # nsmall=32: gathered from level 2 on (every coarse level runs redundantly)
    (4, 1, 16, 32, 8, 16, "FC"),    # 4x1 with nsmall=16: two consecutive gathers (4 -> 2 -> 1 ranks), as bench.py uses for N>1
    (2, 1, 32, 32, 16, 8, "RB"),    # red-black, the default: the reference's sequential order per rank (scan + rank-one correction, mgx_rbseq.hip) to 1e-10
    (2, 2, 32, 32, 16, 32, "RB"),   # the same on 2x2 with every coarse level gathered
    (2, 2, 32, 32, 16, 8, "RB+golden"),  # the reference default against its recorded, decomposition-dependent 2x2 history to 1e-10 -- without rb_exact
    (2, 1, 32, 128, 16, 8, "RB+fuse0"),  # ... with the correction inside the walk's launch on an OPEN level (half-rows of 64 columns, one neighbour): two ranks' fused launches share the card
    (2, 1, 32, 32, 16, 8, "RB+par"),  # the plain parallel sweep (rb_seq = 0): history within 5e-5 of the oracle
    (2, 1, 64, 128, 64, 8, "FC"),   # nz=64: the level-1 kernels of the bench (matrix-free, 3-deep pipeline) with an open side
    (2, 2, 32, 32, 16, 8, "FC+nop2p"), # the exchange callback (torch.distributed) instead of the peer-to-peer pushes
    (2, 2, 16, 16, 8, 8, "FC+bmask"),  # bmask=.true.: masked coefficients + the 4-D cA halo exchange of define_matrix
    (2, 2, 32, 32, 16, 8, "FC+rndtopo"),  # BASELINE config 4's generator (mg_testrndtopo) on the 2x2 decomposition
    (2, 2, 32, 64, 128, 16, "FC"),  # BASELINE config 5's shape: nz=128 (k_relax_tall with open sides) + gather at level 3 (nsmall=16)
    (2, 2, 32, 32, 16, 8, "RB+exact+golden"),  # reference default in the reference's order: its recorded 2x2 history to 1e-10, p bitwise
    (2, 1, 32, 32, 16, 8, "RB+exact"),  # exact-order red-black with one open side and a 2x1 gather
    (2, 2, 32, 32, 16, 8, "FC+connectfail"),  # one rank fails to open its peers' buffers: all ranks fall back to the hooks together, the norms' all-reduce keeps one count on every rank
])
def test_multirank_solve(npx, npy, nx, ny, nz, nsmall, method):
    world, port = npx * npy, _free_port()
    method, _, opt = method.partition("+")  # opt: '+'-separated options of tests/_gpu_rank_worker.py
    args = [str(a) for a in (world, npx, npy, port, nx, ny, nz, nsmall)] + [method] + ([opt] if opt else [])
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gpu_rank_worker.py"), str(r)] + args,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{out[-3000:]}"
        assert f"rank {r} ok" in out


@pytest.mark.parametrize("npx,npy,nx,ny,nz,nsmall", [
    (2, 2, 16, 16, 8, 8),      # the thread-rank machinery on a case the process-per-rank suite covers as well
    (4, 2, 16, 16, 128, 16),   # BASELINE config 5's process grid: 8 ranks, nz = 128, nsmall = 16 -> a 2x2 gather, then a 2x1 gather
    (4, 2, 32, 32, 16, 8),     # 4x2 with the gather on the last level only (2x1 after 2x2 ranks were halved once)
    (4, 2, 64, 64, 16, 32),    # the shape of bench.py --gpus 8: two distributed levels, then 2x1 ranks with one open side, then one rank
])
def test_config5_4x2_ranks_in_one_process(npx, npy, nx, ny, nz, nsmall):
    """BASELINE config 5 asks for a 4x2 decomposition with the coarse-grid gather (nsmall = 16).  The test box
    admits at most 6 processes on its GPU, so the 8 ranks run as 8 THREADS of one process, each on its own libmgx.so instance
    and HIP stream (tests/_gpu_thread_ranks.py): the kernels, neighbour tables (incl. the four corner neighbours), gather
    groups (mg_grids.f90:702-718: 2x2 families, then 2x1) and both transports are the ones a process-per-GPU run uses; only the
    hipIpc mapping of the receive slabs is replaced by plain pointers.  Every rank's p (all levels), b, cA, zr, h bit for bit
    against the oracle's emulated ranks."""
    out = subprocess.run([sys.executable, os.path.join(HERE, "_gpu_thread_ranks.py")] + [str(a) for a in (npx, npy, nx, ny, nz, nsmall)],
                         capture_output=True, text=True, timeout=150)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-3000:]
    for r in range(npx * npy):
        assert f"rank {r} ok" in out.stdout
    if (npx, npy, nz) == (4, 2, 128):
        assert "gathered_levels=[2, 3]" in out.stdout, out.stdout


def test_p2p_timeout_is_agreed_collectively():
    """A rank whose pushes stop for one exchange (test hook "p2p_test_drop"): its neighbours' waits time out, and instead of falling
    back alone (which would leave the others pushing to flags nobody reads) every rank learns it at the next global_sum -- the
    all-reduce carries the health flag -- gets the same error, switches to the hooks, and the repeated solve is bit-identical to
    the oracle; then the pushes come back on collectively (mg_mpi_exchange.f90:504-718 has no counterpart: MPI_Waitall blocks)."""
    out = subprocess.run([sys.executable, os.path.join(HERE, "_gpu_thread_ranks.py"), "2", "2", "16", "16", "8", "8", "drop"],
                         capture_output=True, text=True, timeout=150)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-3000:]
    for r in range(4):
        assert f"rank {r} ok" in out.stdout
    assert out.stderr.count("collective fallback seen") == 4


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` from a bare shell (no launcher): bench.py starts its own ranks before touching the GPU.
    Rehearsed here with both ranks on the one GPU of the box (--backend gloo: host-staged callbacks + hipIpc pushes)."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--size", "64", "64", "16",
                          "--steps", "3", "--warmup", "1", "--sweep-reps", "2", "--nsmall", "8"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["config"]["halo_transport"]
    assert "identical" in (j["config"]["transport_check"] or "")


def test_multirank_long_edge_path(monkeypatch):
    """Exchange kernel with several items per thread (what very long edges use to keep the grid resident), forced here
    through the MGX_P2P_IPT test hook on the 2x2 case with gathers."""
    monkeypatch.setenv("MGX_P2P_IPT", "3")
    test_multirank_solve(2, 2, 32, 32, 16, 8, "FC")


@pytest.mark.parametrize("transport", ["mpi-hooks", "p2p"])
def test_fortran_mpi_harness(tmp_path, transport):
    """The reference's parallel driver shape (Fortran + MPI, fortran/mg_testseamount_gpu_mpi.f90) on 2x2 ranks over
    module nhydro -> libmgx.so with the MPI comm hooks of fortran/mgx_mpi_hooks.cpp; with "p2p" the cycle's halos and
    gathers go through the peer-to-peer pushes instead.  Same answers as the one-rank oracle (four-colour ordering is
    decomposition independent) and as the reference's recorded 2x2 run (tests/golden: 25 iterations)."""
    import re
    import shutil
    import numpy as np
    exe = os.path.join(os.path.dirname(HERE), "fortran", "testseamount_gpu_mpi")
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        pytest.skip("flang or MPI not available when build() ran")
    (tmp_path / "nh_namelist").write_text("&nhparam\n relax_method = 'FC',\n solver_prec = 1.d-10,\n/\n")
    cmd = [mpiexec, "-n", "4", exe, "2", "2", "32", "32", "16"] + (["p2p"] if transport == "p2p" else [])
    env = dict(os.environ, OMP_NUM_THREADS="1")
    import signal
    log = tmp_path / "mpi_run.log"
    with open(log, "w") as f:  # own session + a file instead of pipes: a stuck rank can be killed as a group and cannot block us
        proc = subprocess.Popen(cmd, cwd=tmp_path, stdin=subprocess.DEVNULL, stdout=f, stderr=subprocess.STDOUT, env=env, start_new_session=True)
        try:
            rc = proc.wait(timeout=90)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.wait()
            raise AssertionError("MPI harness timed out:\n" + log.read_text()[-3000:])
    stdout = log.read_text()
    assert rc == 0, stdout[-3000:]
    if transport == "p2p":
        assert "p2p_connected =  1" in stdout
    its = re.findall(r"ite = *(\d+): res = *([0-9.E+-]+) / conv", stdout)
    assert len(its) == 25
    from oracle.mgoracle import make_seamount
    o = make_seamount(64, 64, 16, relax_method="FC", solver_prec=1e-10)
    n, h, _ = o.nhydro_solve()
    for (k, r) in its:
        assert abs(float(r) - h[int(k)]) <= 5.1e-3 * h[int(k)]  # Fortran E10.3: 0.dddE+ee, three significant digits
    sp2 = float(re.search(r"sum_p2 = *([0-9.E+-]+)", stdout).group(1))
    assert np.isclose(sp2, (o.field("p")[1:-1, 1:-1, :] ** 2).sum(), rtol=1e-13)
    o.check_nondivergence()
    sd2 = float(re.search(r"sum_div2 = *([0-9.E+-]+)", stdout).group(1))
    assert np.isclose(sd2, (o.field("b")[1:-1, 1:-1, :] ** 2).sum(), rtol=1e-9)


def test_fortran_mpi_harness_native_rccl(tmp_path):
    """The Fortran + MPI driver with libmgx.so's native RCCL transport instead of the MPI hooks (MPI only broadcasts the
    128-byte id): one rank, because RCCL wants one GPU per rank and the box has one."""
    import re
    import shutil
    exe = os.path.join(os.path.dirname(HERE), "fortran", "testseamount_gpu_mpi")
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        pytest.skip("flang or MPI not available when build() ran")
    (tmp_path / "nh_namelist").write_text("&nhparam\n relax_method = 'FC',\n solver_prec = 1.d-10,\n/\n")
    out = subprocess.run([mpiexec, "-n", "1", exe, "1", "1", "64", "64", "16", "rccl"], cwd=tmp_path, capture_output=True, text=True,
                         timeout=120, env=dict(os.environ, OMP_NUM_THREADS="1"), stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "rccl_connected =  1" in out.stdout and "rccl_selftest_ok =  1" in out.stdout, out.stdout[-2000:]
    assert len(re.findall(r"ite = *(\d+): res = *([0-9.E+-]+) / conv", out.stdout)) == 25
    assert "time spent to solve" in out.stdout and "rescaled performance" in out.stdout  # the summary block of solve_p
