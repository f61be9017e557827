"""Worker for tests/test_parallel_gloo.py: one rank of a gloo world on CPU.  Drives mgroms_amd.parallel.Comm through
the same C function pointers libmgx.so calls, on host buffers, and checks a decomposed halo fill + gather against
the oracle's emulated-MPI result."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, npx, npy, port = (int(a) for a in sys.argv[1:6])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mgroms_amd import nhydro
    from mgroms_amd.parallel import Comm
    from oracle.mgoracle import Oracle

    comm = Comm(device="cpu")
    ex, ar, ag = comm.callbacks()
    nx, ny, nz = 8, 6, 4
    tab = nhydro.level_table(nx, ny, nz, npx, npy, rank, nsmall=8)
    nb = tab[0]["neighb"]

    # ---- halo exchange of a field whose value encodes (owner rank, k, j, i), packed like k_halo_pack
    def fld(r):
        i, j, k = np.meshgrid(np.arange(nx + 2), np.arange(ny + 2), np.arange(nz), indexing="ij")
        a = (1000.0 * r + 100.0 * i + 10.0 * j + k).astype(np.float64)
        a[0], a[-1], a[:, 0], a[:, -1] = -1, -1, -1, -1
        return a
    a = fld(rank)
    send_sl = {0: (slice(1, nx + 1), 1), 1: (nx, slice(1, ny + 1)), 2: (slice(1, nx + 1), ny), 3: (1, slice(1, ny + 1)),
               4: (1, 1), 5: (nx, 1), 6: (nx, ny), 7: (1, ny)}
    recv_sl = {0: (slice(1, nx + 1), 0), 1: (nx + 1, slice(1, ny + 1)), 2: (slice(1, nx + 1), ny + 1), 3: (0, slice(1, ny + 1)),
               4: (0, 0), 5: (nx + 1, 0), 6: (nx + 1, ny + 1), 7: (0, ny + 1)}
    dirs = [d for d in range(8) if nb[d] >= 0]
    sb = [np.ascontiguousarray(a[send_sl[d]]).reshape(-1).copy() for d in dirs]
    rb = [np.empty_like(s) for s in sb]
    n = len(dirs)
    peer = (C.c_int * n)(*[nb[d] for d in dirs])
    cnt = (C.c_int * n)(*[s.size for s in sb])
    sp = (C.c_void_p * n)(*[s.ctypes.data for s in sb])
    rp = (C.c_void_p * n)(*[r.ctypes.data for r in rb])
    for _ in range(3):  # repeated fills: message order between a pair of ranks must stay matched
        assert ex(None, n, peer, sp, rp, cnt) == 0, comm.last_error
    for d, r in zip(dirs, rb):
        a[recv_sl[d]] = r.reshape(a[recv_sl[d]].shape)
    o = Oracle(nx, ny, nz, npx, npy)
    for r in range(world):
        o.field("p", 1, r)[...] = fld(r)
    o.fill_halo(1, "p")
    ref = o.field("p", 1, rank)
    for d in dirs:  # every exchanged edge/corner equals the emulated-MPI result
        assert np.array_equal(a[recv_sl[d]], ref[recv_sl[d]]), (rank, d)

    # ---- all-reduce (global_sum)
    s = np.array([float(rank + 1)])
    assert ar(None, s.ctypes.data, 1) == 0
    assert s[0] == world * (world + 1) / 2

    # ---- all-gather inside the reference's colour groups, for the first gathered level
    tabs = [nhydro.level_table(nx, ny, nz, npx, npy, r, nsmall=8) for r in range(world)]
    glev = next((l for l, d in enumerate(tab) if d["gather"]), None)
    if glev is not None:
        mine = tab[glev]
        members = sorted([r for r in range(world) if tabs[r][glev]["color"] == mine["color"]], key=lambda r: (tabs[r][glev]["key"], r))
        assert len(members) == mine["ngx"] * mine["ngy"]
        cntg = 5
        src = np.full(cntg, float(rank))
        out = np.empty(cntg * len(members))
        grp = (C.c_int * len(members))(*members)
        assert ag(None, grp, len(members), src.ctypes.data, out.ctypes.data, cntg) == 0, comm.last_error
        assert np.array_equal(out, np.repeat(np.array(members, dtype=float), cntg))
        # position in the gathered block = key order (mg_gather.f90:140-171, split uses l=mod(key,2), m=key/2)
        for q, r in enumerate(members):
            k = tabs[r][glev]["key"]
            assert (q % mine["ngx"], q // mine["ngx"]) == (k % 2, k // 2)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
