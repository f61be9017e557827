"""world_size 2 and 4 gloo runs (CPU) of the multi-GPU transport layer: halo exchange, all-reduce and the colour-group
all-gather through the C callbacks of mgroms_amd.parallel.Comm, checked against the oracle's emulated MPI."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("npx,npy", [(2, 1), (1, 2), (2, 2)])
def test_transport_gloo(npx, npy):
    import __graft_entry__ as g
    g.build()
    world, port = npx * npy, _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), str(r), str(world), str(npx), str(npy), str(port)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{out}"
        assert f"rank {r} ok" in out
