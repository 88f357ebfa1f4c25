"""Distributed vector: host mirror of ``acc::Vector`` and the BLAS-1 free
functions of ``src/vector.hpp``.

``Layout`` plays the role of dolfinx's ``IndexMap`` + ``Scatterer`` pair the
reference constructs a vector from (``src/vector.hpp:83-96``): sizes, the packed
send/receive index lists, staging buffers, and the neighbour exchange itself.
The exchange is ``torch.distributed.all_to_all_single`` with per-neighbour split
sizes -- on the ``nccl`` backend that is one grouped ncclSend/ncclRecv over xGMI
(RCCL), issued on RCCL's own stream so that it overlaps the interior-cell kernel
the library enqueues between ``begin`` and ``end`` (``src/laplacian.hpp:378-425``).
The library calls back into :meth:`Layout._exchange` from
``pmg_scatter_fwd_begin/_end``; pack and unpack are HIP kernels.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import call, current_stream, ptr, vp



def _dist():
    import torch.distributed as dist

    return dist


class TorchComm:
    """Exchange and reductions through ``torch.distributed`` (callbacks of the C ABI): the ``nccl``
    backend is RCCL (asynchronous ``all_to_all_single`` on RCCL's stream); any other backend
    (``gloo``) is staged through host memory.  The route for callers that bring their own
    transport; the production route on one node is :class:`RcclComm`."""

    native = None

    def __init__(self, group=None, always_exchange=False, halo="exchange"):
        dist = _dist()
        self.group = group
        self.halo = halo  # "windows": the halo moves through pmg_layout_set_windows, the reductions through this
        self.initialized = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.initialized else 1
        self.rank = dist.get_rank(group) if self.initialized else 0
        # every rank of a multi-rank group takes part in every exchange (a collective), even one that
        # happens to share no dof with anybody (always_exchange: also on a single-rank group, e.g. a
        # rank that is its own periodic neighbour -- used to exercise this branch on one GPU)
        self.distributed = self.world > 1 or (always_exchange and self.initialized)
        self.staged = self.distributed and dist.get_backend(group) != "nccl"

    def splits(self, layout):
        send, recv = [0] * self.world, [0] * self.world
        for r, s, c in zip(layout.neighbors, layout.send_counts, layout.recv_counts):
            send[r] = s
            recv[r] = c
        return send, recv

    def exchange(self, layout, phase):
        dist = _dist()
        if self.staged:
            return self._exchange_staged(layout, phase)
        L = layout
        if phase == 0:  # forward begin: owners' packed values -> ghosts
            L._work = dist.all_to_all_single(L.recv_buffer[: L.n_recv], L.send_buffer[: L.n_send], L._recv_splits,
                                             L._send_splits, group=self.group, async_op=True)
        elif phase == 2:  # reverse begin: ghost values -> owners
            L._work = dist.all_to_all_single(L.send_buffer[: L.n_send], L.recv_buffer[: L.n_recv], L._send_splits,
                                             L._recv_splits, group=self.group, async_op=True)
        elif L._work is not None:  # 1, 3: make the compute stream wait for the arrival
            L._work.wait()
            L._work = None

    def _exchange_staged(self, L, phase):
        """Blocking exchange through host memory for process groups that cannot move device
        buffers (gloo): runs the distributed path on machines without one GPU per rank."""
        import torch

        dist = _dist()
        if phase in (0, 2):
            fwd = phase == 0
            src = L.send_buffer[: L.n_send] if fwd else L.recv_buffer[: L.n_recv]
            dst = L.recv_buffer[: L.n_recv] if fwd else L.send_buffer[: L.n_send]
            insp, outsp = (L._send_splits, L._recv_splits) if fwd else (L._recv_splits, L._send_splits)
            host_in = src.cpu()  # synchronises with the pack kernel on the current stream
            host_out = torch.empty(dst.numel(), dtype=torch.float64)
            dist.all_to_all_single(host_out, host_in, outsp, insp, group=self.group)
            dst.copy_(host_out)

    def gather_objects(self, obj):
        """Set-up time: every rank's picklable ``obj``, in rank order."""
        if not self.initialized:
            return [obj]
        out = [None] * self.world
        _dist().all_gather_object(out, obj, group=self.group)
        return out

    def allreduce(self, layout, host, op):
        import torch

        dist = _dist()
        t = torch.from_numpy(host.copy())
        if dist.get_backend(self.group) == "nccl":
            t = t.to(layout.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM, group=self.group)
        host[:] = t.cpu().numpy()

    def all_to_all_host(self, layout, send, recv_count):
        import torch

        dist = _dist()
        s = torch.from_numpy(np.ascontiguousarray(send))
        r = torch.empty(recv_count, dtype=s.dtype)
        if dist.get_backend(self.group) == "nccl":
            s, r = s.to(layout.device), r.to(layout.device)
        ss, rs = self.splits(layout)
        dist.all_to_all_single(r, s, rs, ss, group=self.group)
        return r.cpu().numpy()


# Identity of THIS process among the ranks that exchange window handles: a random token, not the pid -- ranks in
# different PID namespaces (containers) can share a pid and would then use each other's pointers as their own.
_PROCESS_TOKEN = int.from_bytes(__import__("os").urandom(8), "little")


class RcclComm:
    """The library's native communicator (``pmg_comm``): the neighbour exchange is one group of
    ncclSend/ncclRecv per scatter, the reductions are ncclAllReduce on device scalars, both issued
    by the library with no callback into Python.  One process per GPU; ``unique_id`` from rank 0
    reaches the others through the caller's bootstrap (here: ``torch.distributed``)."""

    def __init__(self, rank: int, size: int, unique_id: bytes, halo="exchange"):
        if len(unique_id) != 128:
            raise ValueError("unique id must be 128 bytes")
        self.rank, self.world = int(rank), int(size)
        self.distributed = True
        self.halo = halo  # "windows": the halo as stores into the neighbours' windows, RCCL for the reductions only
        h = vp()
        call("pmg_comm_create", C.byref(h), self.rank, self.world, C.c_char_p(unique_id))
        self.native = h
        self._host = None

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        call("pmg_comm_unique_id", buf)
        return buf.raw

    @classmethod
    def from_torch(cls, group=None, device=None, halo="exchange"):
        """Bootstrap over an initialised ``torch.distributed`` group (any backend)."""
        import torch

        dist = _dist()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        payload = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(payload, src=0, group=group, device=device)
        c = cls(rank, world, payload[0], halo=halo)
        c._host = TorchComm(group)  # set-up-time host exchanges (numpy arrays) only
        return c

    def size(self) -> int:
        """Ranks of the RCCL communicator itself (``pmg_comm_size``), not of the launcher's group."""
        return int(_lib.lib().pmg_comm_size(self.native))

    def all_to_all_host(self, layout, send, recv_count):
        if self._host is None:
            raise RuntimeError("RcclComm without a host-side bootstrap group cannot move host arrays")
        return self._host.all_to_all_host(layout, send, recv_count)

    def gather_objects(self, obj):
        if self._host is None:
            if self.world == 1:
                return [obj]
            raise RuntimeError("RcclComm without a host-side bootstrap group cannot pass window handles")
        return self._host.gather_objects(obj)

    def __del__(self):
        try:
            if getattr(self, "native", None) is not None:
                _lib.lib().pmg_comm_destroy(self.native)
                self.native = None
        except Exception:
            pass


class WindowComm:
    """The library's communicator WITHOUT a transport library (``pmg_comm_create_windows``): reductions
    and halo as direct stores into windows of device memory the ranks of one node map from each other
    (``hipIpc``), signalled by counters -- two kernels per exchange, no RCCL, no MPI, no host callback.
    The interprocess handles travel once, at set-up, over ``gather`` (any callable returning every
    rank's picklable object in rank order; :meth:`from_torch` uses a ``torch.distributed`` group)."""

    halo = "windows"
    distributed = True
    WINDOW_BYTES = 8 * (2 * 16 + 8 + 2 * 16 * 16384)  # PMG_COMM_WINDOW_BYTES

    def __init__(self, rank: int, size: int, gather, host=None):
        import os

        self.rank, self.world = int(rank), int(size)
        self._gather, self._host = gather, host
        self._window, handle = HaloWindows._alloc(self.WINDOW_BYTES)
        everyone = gather({"handle": handle, "pid": _PROCESS_TOKEN, "pointer": self._window.value})
        self._opened = []
        ptrs = (vp * self.world)()
        for r, info in enumerate(everyone):
            if r == self.rank or info["pid"] == _PROCESS_TOKEN:
                ptrs[r] = info["pointer"]
            else:
                p = vp()
                call("pmg_window_open", C.c_char_p(info["handle"]), C.byref(p))
                self._opened.append(p)
                ptrs[r] = p.value
        h = vp()
        call("pmg_comm_create_windows", C.byref(h), self.rank, self.world, ptrs)
        self.native = h
        gather(True)  # every rank has mapped every window before anybody stores

    @classmethod
    def from_torch(cls, group=None):
        host = TorchComm(group)
        return cls(host.rank, host.world, host.gather_objects, host=host)

    def size(self) -> int:
        return int(_lib.lib().pmg_comm_size(self.native))

    def gather_objects(self, obj):
        return self._gather(obj)

    def all_to_all_host(self, layout, send, recv_count):
        if self._host is None:
            raise RuntimeError("WindowComm without a host-side bootstrap group cannot move host arrays")
        return self._host.all_to_all_host(layout, send, recv_count)

    def __del__(self):
        try:
            lib = _lib.lib()
            if getattr(self, "native", None) is not None:
                lib.pmg_comm_destroy(self.native)
                self.native = None
            for p in getattr(self, "_opened", []):
                lib.pmg_window_close(p)
            self._opened = []
            if getattr(self, "_window", None) is not None:
                lib.pmg_window_free(self._window)
                self._window = None
        except Exception:
            pass


class Layout:
    """Sizes + halo plan of one function space on this rank.

    Parameters mirror the flattened IndexMap/Scatterer of the C ABI:
    ``neighbors`` (ranks, ascending), per-neighbour ``send_counts`` /
    ``recv_counts``, ``send_indices`` (owned positions, grouped by neighbour),
    ``recv_indices`` (ghost positions relative to ``size_local``).  ``comm`` moves the data:
    an :class:`RcclComm` (native), a :class:`TorchComm` or any object with the same
    ``exchange`` / ``allreduce`` methods (callbacks); default: ``TorchComm(group)``.
    """

    def __init__(self, size_local, num_ghosts=0, neighbors=(), send_counts=(), recv_counts=(), send_indices=None,
                 recv_indices=None, group=None, device="cuda", always_exchange=False, comm=None):
        import torch

        self.size_local = int(size_local)
        self.num_ghosts = int(num_ghosts)
        self.neighbors = [int(r) for r in neighbors]
        self.send_counts = [int(c) for c in send_counts]
        self.recv_counts = [int(c) for c in recv_counts]
        self.group = group
        self.device = torch.device(device)
        n_send, n_recv = sum(self.send_counts), sum(self.recv_counts)
        self.n_send, self.n_recv = n_send, n_recv
        si = np.zeros(0, np.int32) if send_indices is None else np.ascontiguousarray(send_indices, dtype=np.int32)
        ri = np.zeros(0, np.int32) if recv_indices is None else np.ascontiguousarray(recv_indices, dtype=np.int32)
        if si.size != n_send or ri.size != n_recv:
            raise ValueError("index list lengths do not match the per-neighbour counts")
        if n_send and (si.min() < 0 or si.max() >= self.size_local):
            raise ValueError("send_indices out of the owned range")
        if n_recv and (ri.min() < 0 or ri.max() >= self.num_ghosts):
            raise ValueError("recv_indices out of the ghost range")
        self.send_indices_host, self.recv_indices_host = si, ri
        self.comm = comm if comm is not None else TorchComm(group, always_exchange)
        self.distributed = bool(self.comm.distributed)
        self._work = None
        self._handle = None
        if self.device.type != "cuda":
            return  # host-only layout (set-up exchanges, CPU tests); no library object
        self.send_indices = torch.from_numpy(si).to(self.device)
        self.recv_indices = torch.from_numpy(ri).to(self.device)
        self.send_buffer = torch.zeros(max(n_send, 1), dtype=torch.float64, device=self.device)
        self.recv_buffer = torch.zeros(max(n_recv, 1), dtype=torch.float64, device=self.device)
        native = getattr(self.comm, "native", None)
        self._staged = bool(getattr(self.comm, "staged", False))
        callbacks = self.distributed and native is None
        if callbacks and hasattr(self.comm, "splits"):
            self._send_splits, self._recv_splits = self.comm.splits(self)
        # callbacks must outlive the handle
        self._cb_exchange = _lib.EXCHANGE_FN(self._exchange) if callbacks else _lib.EXCHANGE_FN()
        self._cb_allreduce = _lib.ALLREDUCE_FN(self._allreduce) if callbacks else _lib.ALLREDUCE_FN()
        h = vp()
        call("pmg_layout_create", C.byref(h), self.size_local, self.num_ghosts, n_send, ptr(self.send_indices),
             ptr(self.send_buffer), n_recv, ptr(self.recv_indices), ptr(self.recv_buffer), self._cb_exchange,
             self._cb_allreduce, vp(0))
        self._handle = h
        if callbacks:
            self._cb_allreduce_max = _lib.ALLREDUCE_FN(lambda user, values, n: self._allreduce(user, values, n, "max"))
            call("pmg_layout_set_allreduce_max", h, self._cb_allreduce_max)
        if native is not None:
            nb = np.ascontiguousarray(self.neighbors, dtype=np.int32)
            sc = np.ascontiguousarray(self.send_counts, dtype=np.int32)
            rc = np.ascontiguousarray(self.recv_counts, dtype=np.int32)
            call("pmg_layout_set_comm", h, native, nb.size, nb.ctypes.data_as(_lib.c_ip),
                 sc.ctypes.data_as(_lib.c_ip), rc.ctypes.data_as(_lib.c_ip))
        self._windows = None
        if self.distributed and getattr(self.comm, "halo", "exchange") == "windows":
            self._windows = HaloWindows(self)
        if hasattr(self.comm, "register"):
            self.comm.register(self)

    # ---- callbacks (invoked from inside pmg_scatter_* / the reductions; torch's current stream
    #      is the stream the library was given) ----
    def _exchange(self, user, phase, stream):
        try:
            self.comm.exchange(self, phase)
            return 0
        except Exception:  # never let an exception cross the C boundary
            import sys
            import traceback

            traceback.print_exc(file=sys.stderr)
            return 1

    def _allreduce(self, user, values, n, op="sum"):
        try:
            host = np.ctypeslib.as_array(values, shape=(n,))
            self.comm.allreduce(self, host, op)
            return 0
        except Exception:
            import sys
            import traceback

            traceback.print_exc(file=sys.stderr)
            return 1

    # ---- host-side forward scatter of a numpy array (set-up and CPU tests) ----
    def scatter_fwd_host(self, x: np.ndarray) -> np.ndarray:
        """Owner -> ghost update of a host array of size_local + num_ghosts entries
        with the same plan, over whatever transport the communicator has for host arrays."""
        if not self.distributed:
            return x
        recv = self.comm.all_to_all_host(self, x[self.send_indices_host], self.n_recv)
        x[self.size_local + self.recv_indices_host] = recv
        return x

    @property
    def handle(self):
        if self._handle is None:
            raise RuntimeError("host-only Layout has no device handle")
        return self._handle

    @property
    def total(self):
        return self.size_local + self.num_ghosts

    def forward_scatters(self) -> int:
        """Forward scatters issued on this layout so far (``pmg_layout_forward_scatters``)."""
        return int(call("pmg_layout_forward_scatters", self.handle))

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().pmg_layout_destroy(self._handle)
                self._handle = None
            if getattr(self, "_windows", None) is not None:
                self._windows.release()
        except Exception:
            pass


class HaloWindows:
    """Window memory of one layout and the neighbours' mapped windows (``pmg_window_*``,
    ``pmg_layout_set_windows``): the 64-byte interprocess handles and the per-neighbour offsets
    travel once, at set-up, as picklable objects over the communicator's bootstrap group."""

    FLAG_WORDS = 4 * 64 + 8

    def __init__(self, layout: "Layout"):
        L = layout
        comm = L.comm
        n = len(L.neighbors)
        sc = np.ascontiguousarray(L.send_counts, dtype=np.int32)
        rc = np.ascontiguousarray(L.recv_counts, dtype=np.int32)
        doubles = C.c_int64()
        fwd, rev = np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int64)
        call("pmg_layout_window_describe", n, sc.ctypes.data_as(_lib.c_ip), rc.ctypes.data_as(_lib.c_ip),
             C.byref(doubles), fwd.ctypes.data_as(_lib.c_lp), rev.ctypes.data_as(_lib.c_lp))
        self.window, wh = self._alloc(8 * doubles.value)
        self.flags, fh = self._alloc(8 * self.FLAG_WORDS)
        self._opened = []
        import os

        # (a window of this very process -- a rank that is its own neighbour, ranks that are threads -- is used
        # through its pointer: a process cannot open its own interprocess handle)
        mine = {"neighbors": list(L.neighbors), "doubles": int(doubles.value), "fwd": fwd[:n].tolist(),
                "rev": rev[:n].tolist(), "window": wh, "flags": fh, "pid": _PROCESS_TOKEN,
                "pointers": (self.window.value, self.flags.value)}
        everyone = comm.gather_objects(mine)
        me = int(comm.rank)
        mapped = {me: (self.window, self.flags)}
        nb_win, nb_flags = (vp * max(n, 1))(), (vp * max(n, 1))()
        nb_doubles, nb_fwd, nb_rev = (np.zeros(max(n, 1), np.int64) for _ in range(3))
        nb_slot = np.zeros(max(n, 1), np.int32)
        seen = {}
        for k, r in enumerate(L.neighbors):
            info = everyone[r]
            # the j-th time rank r appears in my list pairs with the j-th time I appear in r's list
            j = seen.get(r, 0)
            seen[r] = j + 1
            slots = [i for i, q in enumerate(info["neighbors"]) if q == me]
            if j >= len(slots):
                raise ValueError(f"rank {r} does not list rank {me} as a neighbour as often as rank {me} lists it")
            slot = slots[j]
            if r not in mapped:
                if info["pid"] == _PROCESS_TOKEN:
                    mapped[r] = (vp(info["pointers"][0]), vp(info["pointers"][1]))
                else:
                    mapped[r] = (self._open(info["window"]), self._open(info["flags"]))
            nb_win[k], nb_flags[k] = (m.value for m in mapped[r])
            nb_doubles[k], nb_fwd[k], nb_rev[k], nb_slot[k] = info["doubles"], info["fwd"][slot], info["rev"][slot], slot
        call("pmg_layout_set_windows", L.handle, n, sc.ctypes.data_as(_lib.c_ip), rc.ctypes.data_as(_lib.c_ip),
             self.window, self.flags, nb_win, nb_flags, nb_doubles.ctypes.data_as(_lib.c_lp),
             nb_fwd.ctypes.data_as(_lib.c_lp), nb_rev.ctypes.data_as(_lib.c_lp), nb_slot.ctypes.data_as(_lib.c_ip))
        # nobody stores into a window its owner could still free on a failed set-up
        comm.gather_objects(True)

    @staticmethod
    def _alloc(nbytes):
        p, h = vp(), C.create_string_buffer(64)
        call("pmg_window_alloc", nbytes, C.byref(p), h)
        return p, h.raw

    def _open(self, handle):
        p = vp()
        call("pmg_window_open", C.c_char_p(handle), C.byref(p))
        self._opened.append(p)
        return p

    def release(self):
        """After the layout is gone: unmap the neighbours' windows, free mine."""
        lib = _lib.lib()
        for p in self._opened:
            lib.pmg_window_close(p)
        self._opened = []
        for p in (self.window, self.flags):
            if p is not None:
                lib.pmg_window_free(p)
        self.window = self.flags = None


class Vector:
    """``acc::Vector<T, Device::HIP>`` (``src/vector.hpp:74-325``): owned entries
    followed by ghosts, device resident (a torch tensor is the storage)."""

    def __init__(self, layout: Layout, bs: int = 1):
        import torch

        if bs != 1:
            raise ValueError("block size 1 only (the reference hot path uses bs = 1)")
        self.layout = layout
        self._x = torch.zeros(layout.total, dtype=torch.float64, device=layout.device)

    # -- reference API --
    def set(self, v: float):  # :109-115
        call("pmg_vec_set", self.layout.handle, ptr(self._x), float(v), current_stream())

    def copy_from_host(self, other):  # :117-122 (owned part only)
        import torch

        a = np.asarray(other.array() if hasattr(other, "array") and callable(other.array) else other, dtype=np.float64)
        n = self.layout.size_local
        self._x[:n].copy_(torch.from_numpy(np.ascontiguousarray(a[:n])), non_blocking=False)

    def array(self):  # :141-150
        return self._x

    def mutable_array(self):  # :153-162
        return self._x

    def map(self):
        return self.layout

    def bs(self):
        return 1

    def scatter_fwd_begin(self):  # :186-207
        call("pmg_scatter_fwd_begin", self.layout.handle, ptr(self._x), current_stream())

    def scatter_fwd_end(self):  # :209-238
        call("pmg_scatter_fwd_end", self.layout.handle, ptr(self._x), current_stream())

    def scatter_fwd(self):  # :242-246
        self.scatter_fwd_begin()
        self.scatter_fwd_end()

    def scatter_rev_begin(self):  # :249-267
        call("pmg_scatter_rev_begin", self.layout.handle, ptr(self._x), current_stream())

    def scatter_rev_end(self):  # :270-286
        call("pmg_scatter_rev_end", self.layout.handle, ptr(self._x), current_stream())

    def scatter_rev(self):
        self.scatter_rev_begin()
        self.scatter_rev_end()

    def data_copy(self):  # :297-302
        return self._x.cpu().numpy()

    @property
    def data(self):
        return self._x


def _h(v: Vector):
    return v.layout.handle


def inner_product(a: Vector, b: Vector) -> float:  # :334-352
    if a.layout.size_local != b.layout.size_local:
        raise RuntimeError("Incompatible vector sizes")  # :343
    out = C.c_double()
    call("pmg_vec_inner_product", _h(a), ptr(a.data), ptr(b.data), C.byref(out), current_stream())
    return out.value


def squared_norm(a: Vector) -> float:  # :357-362
    return inner_product(a, a)


def norm(a: Vector, kind: str = "l2") -> float:  # :369-390
    if kind not in ("l2", "linf"):
        raise RuntimeError("Norm type not supported")
    out = C.c_double()
    call("pmg_vec_norm", _h(a), ptr(a.data), 0 if kind == "l2" else 1, C.byref(out), current_stream())
    return out.value


def axpy(r: Vector, alpha: float, x: Vector, y: Vector):  # :398-407, r = alpha x + y
    call("pmg_vec_axpy", _h(r), ptr(r.data), float(alpha), ptr(x.data), ptr(y.data), current_stream())


def scale(r: Vector, alpha: float):  # :413-418
    call("pmg_vec_scale", _h(r), ptr(r.data), float(alpha), current_stream())


def copy(a: Vector, b: Vector):  # :424-431, a = b
    call("pmg_vec_copy", _h(a), ptr(a.data), ptr(b.data), current_stream())


def pointwise_mult(w: Vector, x: Vector, y: Vector):  # :438-447
    call("pmg_vec_pointwise_mult", _h(w), ptr(w.data), ptr(x.data), ptr(y.data), current_stream())
