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
    (2, 1, 32, 32, 16, 8, "RB"),    # red-black: parallel semantics, history close to the oracle
    (2, 2, 32, 32, 16, 8, "FC+nop2p"), # the exchange callback (torch.distributed) instead of the peer-to-peer pushes
    (2, 2, 16, 16, 8, 8, "FC+bmask"),  # bmask=.true.: masked coefficients + the 4-D cA halo exchange of define_matrix
])
def test_multirank_solve(npx, npy, nx, ny, nz, nsmall, method):
    world, port = npx * npy, _free_port()
    method, _, opt = method.partition("+")
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
