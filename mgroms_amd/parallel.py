"""Multi-GPU plumbing: the reference's MPI layer (src/mg_mpi_exchange.f90, src/mg_gather.f90) re-expressed as three
callbacks that libmgx.so invokes with DEVICE pointers (include/mgx.h: mgx_set_comm).  One process per GPU;
transport is torch.distributed -- backend "nccl" is RCCL over xGMI on ROCm.

  exchange  <- fill_halo_*: 8-neighbour non-blocking send/recv  (mg_mpi_exchange.f90:504-718)
               one batched group of isend/irecv per halo fill (ncclGroupStart/End underneath): on the fully
               connected xGMI fabric every neighbour is one hop, all edges of a fill travel concurrently.
               With `p2p=True` (default on GPUs) the p/b/r halos of the cycle do not come through this callback at
               all: after nhydro_init the ranks swap hipIpc handles once (connect_p2p) and libmgx.so pushes edges
               straight into the neighbours' buffers over xGMI, flag-synchronised on the device (include/mgx.h,
               "peer-to-peer halo transport").  The callback then only serves the set-up halos.
  allreduce <- global_sum: 1 double                             (mg_mpi_exchange.f90:1555-1571)
  allgather <- gather_3D on the 2x2 / 2x1 colour groups         (mg_gather.f90:126, mg_grids.f90:702-718)
               done with point-to-point messages inside the group (<= 4 members): no sub-communicator needed.

The data path never touches the host with the nccl backend.  With the gloo backend (CPU tests, and several ranks
sharing one GPU on a development box) buffers are staged through host memory; `device="cpu"` lets the same code
run on plain host pointers so the transport logic is testable without a GPU.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from ._lib import ALLGATHER_FN, ALLREDUCE_FN, EXCHANGE_FN


class _DevPtr:
    """Zero-copy torch view of `count` doubles at a raw device pointer (CUDA array interface)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class Comm:
    def __init__(self, device="cuda", group=None, p2p=None, native=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.device = device
        self.p2p = (device == "cuda") if p2p is None else bool(p2p)  # device-to-device halo pushes (one node)
        self.p2p_active = False
        self.p2p_error = None
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.staged = self.backend != "nccl"  # gloo moves host memory only
        # native: libmgx.so's own RCCL communicator serves the three hooks (no Python in the loop); torch.distributed only
        # carries the 128-byte bootstrap id.  Needs one GPU per rank, i.e. the real multi-GPU case (backend nccl).
        # OPT-IN (default off): it has only ever run on a world of one rank (no multi-GPU node was available to this build); until a
        # run on >= 2 GPUs has shown p and the residual history bit-identical through it and through the torch.distributed hooks,
        # gathers included, the hooks carry the set-up halos and the norm and the peer-to-peer pushes the cycle's halos.
        self.native = False if native is None else bool(native)
        self.native_active = False
        self.native_error = None
        self._cache = {}
        self._ops = {}
        self.n_exchange = self.n_allreduce = self.n_allgather = 0
        self._ex = EXCHANGE_FN(self._exchange)
        self._ar = ALLREDUCE_FN(self._allreduce)
        self._ag = ALLGATHER_FN(self._allgather)
        self.last_error = None

    # -- pointer -> tensor ---------------------------------------------------------------------
    def _t(self, ptr, count):
        key = (int(ptr), int(count))
        t = self._cache.get(key)
        if t is None:
            if self.device == "cuda":
                t = torch.as_tensor(_DevPtr(ptr, count), device="cuda")
            else:
                a = np.ctypeslib.as_array(C.cast(C.c_void_p(int(ptr)), C.POINTER(C.c_double)), shape=(int(count),))
                t = torch.from_numpy(a)
            self._cache[key] = t
        return t

    def _p2p(self, sends, recvs, key=None):
        """sends/recvs: lists of (tensor, peer).  One batched group; returns when the data is usable in stream order.
        `key` identifies a recurring exchange (same buffers, peers, counts): its P2POp list is built once."""
        if not sends and not recvs:
            return
        if not self.staged and key is not None:
            ops = self._ops.get(key)
            if ops is None:
                ops = [dist.P2POp(dist.irecv, t, p, self.group) for t, p in recvs] + [dist.P2POp(dist.isend, t, p, self.group) for t, p in sends]
                self._ops[key] = ops
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            return
        if self.staged:
            hs = [(t.detach().to("cpu", copy=True).contiguous(), p) for t, p in sends]
            hr = [(torch.empty(t.shape, dtype=t.dtype), p) for t, p in recvs]
            ops = [dist.P2POp(dist.irecv, t, p, self.group) for t, p in hr] + [dist.P2POp(dist.isend, t, p, self.group) for t, p in hs]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for (dst, _), (src, _) in zip(recvs, hr):
                dst.copy_(src)
        else:
            ops = [dist.P2POp(dist.irecv, t, p, self.group) for t, p in recvs] + [dist.P2POp(dist.isend, t, p, self.group) for t, p in sends]
            for w in dist.batch_isend_irecv(ops):
                w.wait()  # nccl: orders the current stream after the transfer, does not block the host

    # -- callbacks (C ABI) ------------------------------------------------------------------------
    def _exchange(self, ctx, n, peer, sendbuf, recvbuf, count):
        try:
            self.n_exchange += 1
            key = tuple((int(peer[q]), int(sendbuf[q]), int(recvbuf[q]), int(count[q])) for q in range(n))
            if not self.staged and key in self._ops:
                self._p2p(True, True, key)
                return 0
            sends = [(self._t(sendbuf[q], count[q]), int(peer[q])) for q in range(n)]
            recvs = [(self._t(recvbuf[q], count[q]), int(peer[q])) for q in range(n)]
            self._p2p(sends, recvs, key)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            self.last_error = e
            return 1

    def _allreduce(self, ctx, buf, n):
        try:
            self.n_allreduce += 1
            t = self._t(buf, n)
            if self.staged and self.device == "cuda":
                h = t.to("cpu", copy=True)
                dist.all_reduce(h, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, group=self.group)
            return 0
        except Exception as e:
            self.last_error = e
            return 1

    def _allgather(self, ctx, group, ng, sendbuf, recvbuf, count):
        try:
            self.n_allgather += 1
            members = [int(group[q]) for q in range(ng)]
            me = members.index(self.rank)
            src = self._t(sendbuf, count)
            out = self._t(recvbuf, count * ng)
            out[me * count:(me + 1) * count].copy_(src)
            sends = [(src, m) for q, m in enumerate(members) if q != me]
            recvs = [(out[q * count:(q + 1) * count], m) for q, m in enumerate(members) if q != me]
            self._p2p(sends, recvs)
            return 0
        except Exception as e:
            self.last_error = e
            return 1

    # -- wiring -------------------------------------------------------------------------------------
    def callbacks(self):
        return self._ex, self._ar, self._ag

    def install(self):
        """Hand the callbacks to libmgx.so and make it launch on torch's current stream, so that kernels,
        packs/unpacks and the RCCL transfers are ordered on one stream."""
        from ._lib import check, lib
        if self.device == "cuda":
            check(lib().mgx_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        if self.native and (self.native_active or self._connect_native()):
            return
        check(lib().mgx_set_comm(self._ex, self._ar, self._ag, None))

    def _agree(self, flag):
        """all ranks together: True only if `flag` is true on every rank"""
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cpu" if self.staged else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item()) == 1

    def after_init(self):
        """Collective, right after mgx_init on every rank: self-test of the native RCCL transport (a failure puts every
        rank back on the callbacks), then the peer-to-peer halo pushes when asked for.  Every decision is taken by all
        ranks together: a rank that failed alone would otherwise leave its neighbours pushing to flags nobody raises."""
        from ._lib import check, lib
        L = lib()
        if self.native_active:
            ok = L.mgx_rccl_selftest() == 0
            if not self._agree(ok):
                self.native_error = (L.mgx_last_error().decode() if not ok else "self-test failed on another rank")
                L.mgx_rccl_disconnect()
                self.native_active = False
                check(L.mgx_set_comm(self._ex, self._ar, self._ag, None))
        if not self.p2p:
            return
        ok = True
        try:
            self.connect_p2p()
        except Exception as e:  # the halo pushes are an optimisation: anything unexpected leaves the other transport in charge
            ok = False
            self.p2p_error = f"connect_p2p raised {e!r}"
        # the outcome of connect_p2p is already collective when it returns; an exception on one rank is not: agree on it
        if not self._agree(ok):
            self.p2p_active = False
            if self.p2p_error is None:
                self.p2p_error = "connect_p2p failed on another rank"
            L.mgx_set_option(b"p2p", 0)

    def _connect_native(self):
        """Collective: rank 0 creates the RCCL unique id, torch.distributed broadcasts its 128 bytes, every rank joins
        libmgx.so's communicator.  All-or-nothing: if any rank fails, every rank falls back to the callbacks."""
        from ._lib import lib
        L = lib()
        dev = "cpu" if self.staged else "cuda"
        nb = L.mgx_rccl_unique_id_bytes()
        blob = C.create_string_buffer(nb)
        rc = L.mgx_rccl_get_unique_id(blob) if self.rank == 0 else 0
        t = torch.frombuffer(bytearray(blob.raw), dtype=torch.uint8).clone().to(dev)
        src = 0 if self.group is None else dist.get_global_rank(self.group, 0)
        dist.broadcast(t, src=src, group=self.group)
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 1:
            rc = L.mgx_rccl_connect(bytes(t.cpu().numpy().tobytes()), self.world, self.rank)
            ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) != 1:
            self.native_error = L.mgx_last_error().decode() or "a peer could not join the RCCL communicator"
            L.mgx_rccl_disconnect()
            self.native_active = False
            return False
        self.native_active = True
        return True

    def transport(self):
        from ._lib import lib
        return lib().mgx_transport().decode()


    def connect_p2p(self):
        """Collective, after mgx_init: swap the hipIpc handles of the receive slabs and open the neighbours' ones."""
        from ._lib import check, lib
        L = lib()
        nb = L.mgx_p2p_handle_bytes()
        blob = C.create_string_buffer(nb)
        dev = "cpu" if self.staged else "cuda"

        def all_ok(flag):  # every step is all-or-nothing: a rank that fails must not leave the others in a collective
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            return int(t.item()) == 1

        def give_up(msg):
            self.p2p_error = msg
            L.mgx_set_option(b"p2p", 0)
            self.p2p_active = False
            return False

        rc = L.mgx_p2p_prepare(blob)
        if not all_ok(rc == 0):
            return give_up(L.mgx_last_error().decode() if rc else "a peer could not export its halo buffers (hipIpcGetMemHandle)")
        mine = torch.frombuffer(bytearray(blob.raw), dtype=torch.uint8).clone().to(dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        allh = b"".join(bytes(t.cpu().numpy().tobytes()) for t in parts)
        rc = L.mgx_p2p_connect(allh, self.world)
        if not all_ok(rc == 0):
            return give_up(L.mgx_last_error().decode() if rc else "a peer could not open the shared halo buffers (hipIpcOpenMemHandle)")
        self.p2p_active = True
        return self._p2p_selftest()

    def _p2p_selftest(self):
        """One level-1 halo fill of a rank-coded field through both transports; the peer-to-peer one stays on only if
        every rank got exactly the halos the callback delivers (a dead peer shows up as the 5 s device time-out)."""
        from ._lib import MgxError, lib
        from . import nhydro
        L = lib()
        g = nhydro.grid(1)
        rng = np.random.default_rng(1000 + self.rank)
        pat = rng.standard_normal(g._shape("p"))
        good = 1
        try:
            g.set("p", pat)
            nhydro.fill_halo(1, "p")
            a = g.get("p")
        except MgxError as e:
            self.p2p_error = str(e)
            good, a = 0, None
        L.mgx_set_option(b"p2p", 0)
        g.set("p", pat)
        nhydro.fill_halo(1, "p")
        if good and not np.array_equal(a, g.get("p")):
            good, self.p2p_error = 0, "self-test: halos differ from the callback transport"
        g.set("p", np.zeros_like(pat))
        ok = torch.tensor([good], dtype=torch.int32, device="cpu" if self.staged else "cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        self.p2p_active = int(ok.item()) == 1
        if self.p2p_active:
            L.mgx_set_option(b"p2p", 1)
        elif self.p2p_error is None:
            self.p2p_error = "self-test failed on another rank"
        return self.p2p_active

    def set_p2p(self, on):
        """Switch between the peer-to-peer transport and the exchange callback (all ranks together)."""
        from ._lib import check, lib
        check(lib().mgx_set_option(b"p2p", 1 if on else 0))
        self.p2p_active = bool(on)


class ThreadWorld:
    """Shared mailbox of `world` ranks that run as THREADS of one process, each driving its own libmgx.so instance
    (include/mgx.h: mgx_instance_*) on its own HIP stream of the one device.  What MPI_COMM_WORLD is to the reference's ranks."""

    def __init__(self, world):
        import queue
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.q = {(a, b): queue.Queue() for a in range(world) for b in range(world)}    # messages a -> b
        self.ack = {(a, b): queue.Queue() for a in range(world) for b in range(world)}  # b has copied a's message
        self.scal = [None] * world
        self.slab = [None] * world
        self.flags = [None] * world
        self.votes = [1] * world


class ThreadComm:
    """mgx_set_comm hooks between thread-ranks of one process (the reference's MPI layer, mg_mpi_exchange.f90:504-718,1555-1571,
    mg_gather.f90:126, as device-to-device copies): a sender synchronises its stream and posts the device pointer, the receiver
    copies on its own stream and acknowledges.  Same interface as Comm (install / after_init / set_p2p / transport)."""
    TIMEOUT = 40.0

    def __init__(self, tw, rank, p2p=True):
        self.tw, self.rank, self.world, self.p2p = tw, rank, tw.world, bool(p2p)
        self.p2p_active, self.p2p_error = False, None
        self.native_active, self.native_error = False, "thread ranks share one device"
        self.last_error = None
        self._ex, self._ar, self._ag = EXCHANGE_FN(self._exchange), ALLREDUCE_FN(self._allreduce), ALLGATHER_FN(self._allgather)

    @staticmethod
    def _t(ptr, count):
        return torch.as_tensor(_DevPtr(ptr, count), device="cuda")

    def _move(self, sends, recvs):
        """sends: [(ptr, count, peer)], recvs: [(ptr, count, peer)]"""
        torch.cuda.current_stream().synchronize()          # my send buffers are complete
        for ptr, cnt, peer in sends:
            self.tw.q[(self.rank, peer)].put((int(ptr), int(cnt)))
        for ptr, cnt, peer in recvs:
            sp, sc = self.tw.q[(peer, self.rank)].get(timeout=self.TIMEOUT)
            if sc != cnt:
                raise RuntimeError(f"rank {self.rank}: message of {sc} doubles from rank {peer}, expected {cnt}")
            self._t(ptr, cnt).copy_(self._t(sp, sc))
        torch.cuda.current_stream().synchronize()          # the copies are done: the senders may reuse their buffers
        for _, _, peer in recvs:
            self.tw.ack[(peer, self.rank)].put(1)
        for _, _, peer in sends:
            self.tw.ack[(self.rank, peer)].get(timeout=self.TIMEOUT)

    def _exchange(self, ctx, n, peer, sendbuf, recvbuf, count):
        try:
            self._move([(sendbuf[q], count[q], int(peer[q])) for q in range(n)], [(recvbuf[q], count[q], int(peer[q])) for q in range(n)])
            return 0
        except Exception as e:
            self.last_error = e
            return 1

    def _allreduce(self, ctx, buf, n):
        try:
            t = self._t(buf, n)
            self.tw.scal[self.rank] = t.cpu().numpy().copy()
            self.tw.barrier.wait(self.TIMEOUT)
            tot = self.tw.scal[0].copy()
            for r in range(1, self.world):             # rank order on every rank: the same bits everywhere
                tot = tot + self.tw.scal[r]
            self.tw.barrier.wait(self.TIMEOUT)
            t.copy_(torch.from_numpy(tot))
            torch.cuda.current_stream().synchronize()
            return 0
        except Exception as e:
            self.last_error = e
            return 1

    def _allgather(self, ctx, group, ng, sendbuf, recvbuf, count):
        try:
            members = [int(group[q]) for q in range(ng)]
            me = members.index(self.rank)
            self._t(int(recvbuf) + 8 * me * count, count).copy_(self._t(sendbuf, count))
            self._move([(sendbuf, count, m) for q, m in enumerate(members) if q != me],
                       [(int(recvbuf) + 8 * q * count, count, m) for q, m in enumerate(members) if q != me])
            return 0
        except Exception as e:
            self.last_error = e
            return 1

    def install(self):
        from ._lib import check, lib
        check(lib().mgx_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        check(lib().mgx_set_comm(self._ex, self._ar, self._ag, None))

    def _agree(self, flag):
        self.tw.votes[self.rank] = 1 if flag else 0
        self.tw.barrier.wait(self.TIMEOUT)
        ok = all(self.tw.votes)
        self.tw.barrier.wait(self.TIMEOUT)
        return ok

    def after_init(self):
        """collective over the thread-ranks, after mgx_init: connect the peer-to-peer pushes through plain pointers"""
        from ._lib import lib
        if not self.p2p:
            return
        L = lib()
        blob = C.create_string_buffer(L.mgx_p2p_handle_bytes())
        slab, flags = C.c_void_p(), C.c_void_p()
        rc = L.mgx_p2p_prepare(blob) or L.mgx_p2p_local_pointers(C.byref(slab), C.byref(flags))
        self.tw.slab[self.rank], self.tw.flags[self.rank] = slab.value, flags.value
        if not self._agree(rc == 0):
            self.p2p_error = L.mgx_last_error().decode() if rc else "a peer could not allocate its halo buffers"
            return
        a = (C.c_void_p * self.world)(*self.tw.slab)
        b = (C.c_void_p * self.world)(*self.tw.flags)
        rc = L.mgx_p2p_connect_pointers(a, b, self.world)
        self.p2p_active = self._agree(rc == 0)
        if not self.p2p_active:
            self.p2p_error = L.mgx_last_error().decode() if rc else "a peer could not connect"
            L.mgx_set_option(b"p2p", 0)

    def set_p2p(self, on):
        from ._lib import check, lib
        self.tw.barrier.wait(self.TIMEOUT)   # between exchanges, all ranks together
        check(lib().mgx_set_option(b"p2p", 1 if on else 0))
        self.p2p_active = bool(on)
        self.tw.barrier.wait(self.TIMEOUT)

    def transport(self):
        from ._lib import lib
        return lib().mgx_transport().decode()


def process_grid(world):
    """npx x npy used by bench.py for 1/2/4/8 GPUs (the reference's power-of-two cartesian grid, assumptions:1-6)."""
    return {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2), 16: (4, 4)}[world]
