"""One rank per GPU: the particle cloud sharded by owning cell range, mesh replicated.

The reference drives ONE GPU from the MPI master rank (src/advect.H:59-89); the MI355X-native scale-out the north star
asks for lives BEHIND THE C-ABI, in cudaparticlesfoam_amd/csrc/cpf_shard*.{h,cpp} and cpf_comm.cpp: rank r
owns the cells ``[cell_lo[r], cell_lo[r+1])``; after a step, particles whose cell belongs to another rank are packed by
the HIP hand-off kernels (ballot/prefix compaction) and exchanged with ONE variable-size all-to-all over RCCL (grouped
ncclSend/ncclRecv: xGMI is a full point-to-point mesh, so every pair's traffic takes its own link), on a side stream
while the step loop runs on.

This module is a BINDING of those calls (``cpf_comm_*``, ``cpf_shard_*``, include/cpf.h) -- the one implementation an
OpenFOAM solver linked against the library uses too (compat/src/initCuda.H) -- plus the host-side helpers that prepare
a case (cell renumbering into x-slabs, initial cuts).

Because the mesh and U are replicated, ownership is organisational, not a correctness requirement: a rank can step any
particle.  The hand-off cadence (``exchange_interval``) is therefore decoupled from the step cadence.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib as L


def slab_cell_ranges(weights: np.ndarray, n_ranks: int) -> np.ndarray:
    """Split cells 0..nC-1 (already numbered so that consecutive ids are spatially coherent) into
    ``n_ranks`` contiguous ranges of equal total weight (e.g. cell volume ~ expected particle count).
    Returns cell_lo[n_ranks+1] (int32)."""
    w = np.asarray(weights, dtype=np.float64)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    targets = cum[-1] * np.arange(1, n_ranks) / n_ranks
    cuts = np.searchsorted(cum, targets, side="left")
    lo = np.concatenate([[0], cuts, [w.size]]).astype(np.int32)
    for r in range(1, n_ranks + 1):          # keep ranges non-decreasing even for degenerate weights
        lo[r] = max(lo[r], lo[r - 1])
    return lo


def x_slab_renumbering(centres: np.ndarray) -> np.ndarray:
    """new_of_old for a cell numbering sorted by (x, y, z) of the cell centre: ranks then own
    contiguous x-slabs (blockMesh numbering is block-major)."""
    order = np.lexsort((centres[:, 2], centres[:, 1], centres[:, 0]))
    new_of_old = np.empty(order.size, dtype=np.int64)
    new_of_old[order] = np.arange(order.size)
    return new_of_old


def slab_bounding_box(mesh, cell_first: int, cell_last: int):
    """Axis-aligned bounding box (lower, upper) of the cells [cell_first, cell_last): where a rank
    samples when it seeds its own slab."""
    own = np.asarray(mesh.owner); nei = np.asarray(mesh.neighbour)
    fo = np.asarray(mesh.face_offsets)
    sel = (own >= cell_first) & (own < cell_last)
    sel[: nei.size] |= (nei >= cell_first) & (nei < cell_last)
    faces = np.nonzero(sel)[0]
    if faces.size == 0:
        raise ValueError("empty cell range [%d, %d)" % (cell_first, cell_last))
    nv = fo[faces + 1] - fo[faces]
    idx = np.repeat(fo[faces], nv) + (np.arange(int(nv.sum())) - np.repeat(np.cumsum(nv) - nv, nv))
    pts = np.asarray(mesh.points)[np.asarray(mesh.face_verts)[idx]]
    return pts.min(0), pts.max(0)


# ---------------------------------------------------------------------------------------------------------------------
# communicators (include/cpf.h "cpf_comm")
# ---------------------------------------------------------------------------------------------------------------------
def unique_id(kind: int = 0) -> bytes:
    """The rendezvous token of a new communicator (cpf_comm_unique_id): ONE rank calls this and the host broadcasts the
    bytes (torch.distributed.broadcast_object_list, Pstream::scatter, MPI_Bcast ...).  kind: L.COMM_RCCL,
    L.COMM_INPROCESS (ranks = threads of this process) or 0 (environment variable CPF_COMM, else RCCL)."""
    lib = L.load()
    buf = C.create_string_buffer(L.COMM_ID_BYTES)
    st = lib.cpf_comm_unique_id(buf, int(kind))
    if st != L.CPF_OK:
        raise L.CpfError(st, (lib.cpf_comm_last_error() or b"").decode())
    return buf.raw


class Communicator:
    """A ``cpf_comm`` made by the library (RCCL, or in-process between threads)."""

    def __init__(self, comm_id: bytes, rank: int, world: int, device: int = 0):
        self.lib = L.load()
        out = C.POINTER(L.Comm)()
        st = self.lib.cpf_comm_create(C.c_char_p(comm_id), int(rank), int(world), int(device), C.byref(out))
        if st != L.CPF_OK:
            raise L.CpfError(st, (self.lib.cpf_comm_last_error() or b"").decode())
        self.ptr, self.rank, self.world = out, int(rank), int(world)
        self._shards = []               # shards that borrow this communicator: destroyed first

    def close(self):
        for sh in list(getattr(self, "_shards", [])):
            sh.close()
        if getattr(self, "ptr", None):
            self.lib.cpf_comm_destroy(self.ptr)
            self.ptr = None


class ShardedCloud:
    """This rank's shard (``cpf_shard``): SoA device arrays with slack capacity + hand-off buffers, all owned by the library.

    ``ctx``: an ``api.Context`` with mesh and velocity set (or, in the CPU tests, the stand-in that tests/host_shard
    provides).  ``comm``: ``None`` (one rank), a ``Communicator``, or anything with ``.ptr`` (a ``POINTER(L.Comm)``),
    ``.rank`` and ``.world``."""

    def __init__(self, ctx, cell_lo: Optional[Sequence[int]], capacity: int, comm=None, send_fraction: float = 0.25,
                 exchange_interval: int = 1, lib=None):
        self.lib = lib if lib is not None else L.load()
        self.ctx, self.comm = ctx, comm
        self.rank = 0 if comm is None else int(comm.rank)
        self.world = 1 if comm is None else int(comm.world)
        lo = None
        if cell_lo is not None:
            lo = np.ascontiguousarray(cell_lo, dtype=np.int32)
            assert lo.size == self.world + 1
        h = C.c_void_p()
        st = self.lib.cpf_shard_create(ctx.h, None if comm is None else comm.ptr, int(capacity),
                                       None if lo is None else lo.ctypes.data_as(C.c_void_p), C.byref(h))
        if st != L.CPF_OK:
            raise L.CpfError(st, (self.lib.cpf_shard_last_error(None) or b"").decode())
        self.h = h
        # the shard borrows the context and the communicator (include/cpf.h): whoever closes one of them first closes the shard first
        for owner in (ctx, comm):
            if owner is not None and hasattr(owner, "__dict__"):
                owner.__dict__.setdefault("_shards", []).append(self)
        self._opts = dict(exchange_interval=0, rebalance_interval=0, overlap_steps=0, sort_interval=0, balance_by_time=0,
                          force_collectives=0, profile_comm=0, send_fraction=0.25)
        self.send_fraction = send_fraction
        self.exchange_interval = exchange_interval

    # -- plumbing
    def _ck(self, st):
        if st != L.CPF_OK:
            raise L.CpfError(st, (self.lib.cpf_shard_last_error(self.h) or b"").decode("utf-8", "replace"))

    def close(self):
        if getattr(self, "h", None):
            self.lib.cpf_shard_destroy(self.h)
            self.h = None
            for owner in (self.ctx, self.comm):
                lst = getattr(owner, "_shards", None)
                if lst is not None and self in lst:
                    lst.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key: str, value: float):
        self._ck(self.lib.cpf_shard_set_option(self.h, key.encode(), float(value)))
        self._opts[key] = value

    def stats(self) -> L.ShardStats:
        s = L.ShardStats()
        self._ck(self.lib.cpf_shard_get_stats(self.h, C.byref(s)))
        return s

    # options as attributes (set the same on every rank)
    exchange_interval = property(lambda s: int(s._opts["exchange_interval"]), lambda s, v: s.set_option("exchange_interval", max(0, int(v))))
    rebalance_interval = property(lambda s: int(s._opts["rebalance_interval"]), lambda s, v: s.set_option("rebalance_interval", int(v)))
    overlap_steps = property(lambda s: int(s._opts["overlap_steps"]), lambda s, v: s.set_option("overlap_steps", int(v)))
    sort_interval = property(lambda s: int(s._opts["sort_interval"]), lambda s, v: s.set_option("sort_interval", int(v)))
    force_collectives = property(lambda s: bool(s._opts["force_collectives"]), lambda s, v: s.set_option("force_collectives", 1 if v else 0))
    profile_comm = property(lambda s: bool(s._opts["profile_comm"]), lambda s, v: s.set_option("profile_comm", 1 if v else 0))
    send_fraction = property(lambda s: float(s._opts["send_fraction"]), lambda s, v: s.set_option("send_fraction", float(v)))
    balance_by_time = property(lambda s: bool(s._opts["balance_by_time"]))

    def enable_time_balancing(self, on: bool = True):
        """Re-cut by measured cost instead of by particle count (must be set the same on every rank)."""
        self.set_option("balance_by_time", 1 if on else 0)

    # counters
    n = property(lambda s: int(s.stats().n))
    capacity = property(lambda s: int(s.stats().capacity))
    particle_steps = property(lambda s: int(s.stats().particleSteps))
    handed_off = property(lambda s: int(s.stats().handedOff))
    exchanges = property(lambda s: int(s.stats().exchanges))
    rebalances = property(lambda s: int(s.stats().rebalances))
    grown = property(lambda s: int(s.stats().grown))
    send_grown = property(lambda s: int(s.stats().sendGrown))
    kernel_ms = property(lambda s: float(s.stats().kernelMs))
    kernel_launches = property(lambda s: int(s.stats().kernelLaunches))
    handoff_host_ms = property(lambda s: float(s.stats().handoffHostMs))
    handoff_wait_ms = property(lambda s: float(s.stats().handoffWaitMs))
    step_index = property(lambda s: int(s.stats().stepIndex), lambda s, v: s.set_option("step_index", int(v)))

    def comm_ms(self) -> float:
        """Device time of the hand-off collectives so far (counts all-gather + payload all-to-all, side stream; needs
        ``profile_comm``)."""
        return float(self.stats().commDeviceMs)

    def _overlap(self) -> int:
        return int(self.stats().overlapDepth)

    @property
    def cell_lo(self) -> np.ndarray:
        lo = np.empty(self.world + 1, np.int32)
        self._ck(self.lib.cpf_shard_cell_ranges(self.h, lo.ctypes.data_as(C.c_void_p)))
        return lo

    # -- filling
    @staticmethod
    def _p(t):
        """raw address of a torch tensor / numpy array / int / None"""
        if t is None:
            return None
        if isinstance(t, int):
            return C.c_void_p(t)
        if hasattr(t, "data_ptr"):
            return C.c_void_p(t.data_ptr())
        return t.ctypes.data_as(C.c_void_p)

    def set_particles(self, x, y, z, cell, gid, first_gid: int = 0, n: Optional[int] = None):
        """x, y, z, cell (None: located here), gid (None: first_gid, first_gid + 1, ...): arrays in DEVICE memory (torch
        tensors, or raw addresses with ``n``) of equal length; they are copied."""
        if n is None:
            n = int(x.numel()) if hasattr(x, "numel") else int(np.asarray(x).size)
        self._ck(self.lib.cpf_shard_set_particles_dev(self.h, self._p(x), self._p(y), self._p(z), self._p(cell), self._p(gid), n,
                                                      int(first_gid)))

    def seed_box(self, n_total: int, lower, upper, order: int = 1) -> int:
        """cudaInitParticles + locate + hand-off to the owners for the WHOLE cloud (collective); returns the number of
        out-of-domain particles."""
        lo = np.ascontiguousarray(lower, dtype=np.float64); hi = np.ascontiguousarray(upper, dtype=np.float64)
        out = C.c_int64()
        self._ck(self.lib.cpf_shard_seed_box(self.h, int(n_total), lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p),
                                             int(order), C.byref(out)))
        return out.value

    def set_velocity(self, U):
        """New cell velocities (transient solvers, src/advect.H:44-57).  A hand-off still in flight is completed FIRST: its
        arrivals replay the cycles they missed in one launch, and that replay must see the field those cycles were stepped
        with, not the new one."""
        U = np.ascontiguousarray(U, dtype=np.float64)
        self._ck(self.lib.cpf_shard_set_velocity(self.h, U.ctypes.data_as(C.c_void_p), U.shape[0]))

    def set_velocity_slice(self, U_slice):
        """This rank's slice of the field (the cells of its piece of the decomposed mesh); all-gathered between the GPUs."""
        U = np.ascontiguousarray(U_slice, dtype=np.float64).reshape(-1, 3)
        self._ck(self.lib.cpf_shard_set_velocity_slice(self.h, U.ctypes.data_as(C.c_void_p), U.shape[0]))

    # -- the hot loop
    def step(self, dt: float, n_cycles: int = 1, D: float = 0.0, flags: int = 0):
        self._ck(self.lib.cpf_shard_step(self.h, float(dt), float(D), int(n_cycles), int(flags)))

    def flush(self):
        """Completes a hand-off still in flight (arrivals appended and caught up)."""
        self._ck(self.lib.cpf_shard_flush(self.h))

    def exchange(self):
        """Hand particles that left this rank's cell range to their owners (all-to-all-v), synchronously."""
        self._ck(self.lib.cpf_shard_exchange(self.h))

    def rebalance(self, n_cells: Optional[int] = None):
        """Re-cut the ranges to equal particle counts (or equal measured cost) and hand particles to their new owners."""
        self._ck(self.lib.cpf_shard_rebalance(self.h))

    def sort(self):
        self._ck(self.lib.cpf_shard_sort(self.h))

    # -- inspection
    def arrays(self):
        """Device addresses of the shard's arrays and its size: dict(x, y, z, cell, gid, n, capacity); valid until the next
        call that may grow, sort or exchange."""
        p = [C.c_void_p() for _ in range(5)]
        n, cap = C.c_int64(), C.c_int64()
        self._ck(self.lib.cpf_shard_arrays(self.h, *[C.byref(q) for q in p], C.byref(n), C.byref(cap)))
        return dict(x=p[0].value, y=p[1].value, z=p[2].value, cell=p[3].value, gid=p[4].value, n=n.value, capacity=cap.value)

    def gather_to_numpy(self):
        """(gid, x, y, z, cell) of this shard on the host."""
        n = self.arrays()["n"]
        g = np.empty(n, np.int64); x = np.empty(n); y = np.empty(n); z = np.empty(n); c = np.empty(n, np.int32)
        self._ck(self.lib.cpf_shard_get_local(self.h, *[a.ctypes.data_as(C.c_void_p) for a in (g, x, y, z, c)]))
        return g, x, y, z, c

    def global_count(self) -> int:
        out = C.c_int64()
        self._ck(self.lib.cpf_shard_global_count(self.h, C.byref(out)))
        return out.value

    def gather(self, root: int = 0, want_vel: bool = False):
        """COLLECTIVE: the whole cloud in particle-id order on ``root`` -- (xyzw [n][4], cell [n], vel [n][4] or None);
        (None, None, None) elsewhere."""
        total = self.global_count()
        if self.rank != root:
            self._ck(self.lib.cpf_shard_gather(self.h, int(root), None, None, None))
            return None, None, None
        xyzw = np.empty((total, 4)); cell = np.empty(total, np.int32); vel = np.empty((total, 4)) if want_vel else None
        self._ck(self.lib.cpf_shard_gather(self.h, int(root), xyzw.ctypes.data_as(C.c_void_p), cell.ctypes.data_as(C.c_void_p),
                                           None if vel is None else vel.ctypes.data_as(C.c_void_p)))
        return xyzw, cell, vel

    def write_vtu(self, path: str, root: int = 0, want_ke: bool = True):
        """COLLECTIVE: one frame of the whole cloud written by ``root`` (the reference's particle_%04d.vtu layout): gathered to
        the root's GPU, copied, summed, formatted and written by its worker thread.  want_ke=False: returns None after the
        device-side gather; True: the root waits for the copy and returns the total kinetic energy (0.0 elsewhere)."""
        ke = C.c_double()
        st = self.lib.cpf_shard_write_vtu(self.h, int(root), path.encode(), C.byref(ke) if want_ke else None)
        if st not in (L.CPF_OK, L.CPF_WARN_NAN):
            self._ck(st)
        return ke.value if want_ke else None
