"""One process per GPU: the particle cloud sharded by owning cell range, mesh replicated.

The reference drives ONE GPU from the MPI master rank (src/advect.H:59-89); this module is the
MI355X-native scale-out the north star asks for: rank r owns the cells
``[cell_lo[r], cell_lo[r+1])``; after a step, particles whose cell belongs to another rank are
packed by the HIP hand-off kernels (ballot/prefix compaction) and exchanged with ONE
variable-size all-to-all over RCCL (``torch.distributed`` backend "nccl" == RCCL; xGMI is a
full point-to-point mesh, so an all-to-all-v uses each link once with only that pair's
traffic).  torch is plumbing here: device memory, streams and the process group.

Because the mesh and U are replicated, ownership is organisational, not a correctness
requirement: a rank can step any particle.  The hand-off cadence (``exchange_interval``) is
therefore decoupled from the step cadence.
"""
from __future__ import annotations

import contextlib
import time
from typing import Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

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


COST_UNIT_MS = 2.0e-8      # ms per particle-step that counts as cost 1 (~ the measured single-GPU rate)


def device_cell_ranges(hist: torch.Tensor, n_ranks: int) -> torch.Tensor:
    """``slab_cell_ranges`` on the histogram's own device (int64 counts or float64 weights -> int32
    cell_lo[n_ranks+1]); counts are exact in float64 below 2^53, so both give the same cuts."""
    cum = torch.cumsum(hist.to(torch.float64), 0)
    cum0 = torch.cat([cum.new_zeros(1), cum])
    targets = cum[-1] * torch.arange(1, n_ranks, dtype=torch.float64, device=hist.device) / n_ranks
    cuts = torch.searchsorted(cum0, targets, right=False)
    lo = torch.cat([cuts.new_zeros(1), cuts, cuts.new_full((1,), hist.numel())])
    return torch.cummax(lo, 0).values.to(torch.int32)


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


class HipOps:
    """Device operations of a shard, all through the C-ABI (no CPU fallback)."""

    def __init__(self, ctx):
        self.ctx = ctx
        # ShardedCloud mixes torch operations (copies, cat, collectives, reallocation) with the context's kernels:
        # they are only ordered if both run on the SAME stream, so the context is put on torch's current stream
        # here instead of leaving that to the caller
        if torch.cuda.is_available():
            ctx.set_stream(torch.cuda.current_stream(torch.device("cuda", ctx.device)).cuda_stream)

    def set_velocity(self, U):
        self.ctx.set_velocity(U)

    @staticmethod
    def _p(t: Optional[torch.Tensor]):
        return None if t is None else t.data_ptr()

    def step(self, s: "ShardedCloud", dt, D, step0, n_cycles, flags):
        self.ctx.step_dev(self._p(s.x), self._p(s.y), self._p(s.z), self._p(s.cell), self._p(s.gid), None, s.n, dt, D,
                          step0, n_cycles, flags)

    def step_slice(self, s: "ShardedCloud", first: int, count: int, dt, D, step0, n_cycles, flags):
        """Steps only particles [first, first+count) for n_cycles (one fused launch): arrivals catching up on the
        steps they missed while in flight.  Not a timed launch (keeps the balancer's per-launch times clean)."""
        timing = s.balance_by_time or s.timing_on
        if timing:
            self.ctx.timing_enable(False)
        self.ctx.step_dev(s.x.data_ptr() + 8 * first, s.y.data_ptr() + 8 * first, s.z.data_ptr() + 8 * first,
                          s.cell.data_ptr() + 4 * first, s.gid.data_ptr() + 8 * first, None, count, dt, D, step0, n_cycles,
                          flags | L.STEP_FUSE_CYCLES)
        if timing:
            self.ctx.timing_enable(True)

    def pack(self, s: "ShardedCloud"):
        self.ctx.pack_leavers_dev(self._p(s.x), self._p(s.y), self._p(s.z), self._p(s.cell), self._p(s.gid), s.n,
                                  self._p(s.cell_lo_dev), s.world, s.rank, self._p(s.sendbuf), s.send_capacity,
                                  self._p(s.counts_dev), self._p(s.nstay_dev))

    def unpack(self, s: "ShardedCloud", n_stay: int, recvbuf: torch.Tensor, n_recv: int):
        self.ctx.unpack_arrivals_dev(self._p(s.x), self._p(s.y), self._p(s.z), self._p(s.cell), self._p(s.gid), n_stay,
                                     self._p(recvbuf), n_recv)

    def enable_timing(self, s: "ShardedCloud"):
        self.ctx.timing_enable(True)

    def step_time(self, s: "ShardedCloud", wait: bool):
        """(launches, summed device ms) of this rank's step launches since the last call; ``wait`` blocks for
        the launches still in flight, otherwise only finished ones are drained."""
        return self.ctx.timing_read() if wait else self.ctx.timing_poll()

    def histogram(self, s: "ShardedCloud", scale: float):
        self.ctx.cell_histogram_dev(self._p(s.cell), s.n, scale, self._p(s.weights_dev))

    def cell_ranges(self, s: "ShardedCloud"):
        self.ctx.cell_ranges_dev(self._p(s.weights_dev), s.world, self._p(s.cell_lo_dev))

    def sort(self, s: "ShardedCloud"):
        """Into the shard's second set of arrays, which then swap roles with the first (no staging, no copy back: a
        fifth of the sort's time; 36 B per particle slot more memory)."""
        if s.n <= 1:
            return
        alt = getattr(s, "_alt", None)
        if alt is None or alt["x"].numel() != s.capacity:
            alt = {name: torch.empty_like(getattr(s, name)) for name in ("x", "y", "z", "gid")}
            alt["cell"] = torch.full_like(s.cell, L.CELL_LOST)
            s._alt = alt
        self.ctx.sort_by_cell_dev_to(self._p(s.x), self._p(s.y), self._p(s.z), self._p(s.cell), self._p(s.gid),
                                     self._p(alt["x"]), self._p(alt["y"]), self._p(alt["z"]), self._p(alt["cell"]),
                                     self._p(alt["gid"]), s.n)
        for name in ("x", "y", "z", "cell", "gid"):
            cur = getattr(s, name)
            setattr(s, name, alt[name]); alt[name] = cur

    def locate(self, s: "ShardedCloud"):
        self.ctx.locate_initial_dev(self._p(s.x), self._p(s.y), self._p(s.z), self._p(s.cell), s.n)


class ShardedCloud:
    """This rank's shard: SoA device arrays with slack capacity + hand-off buffers."""

    def __init__(self, ops, cell_lo: Sequence[int], capacity: int, device: torch.device, rank: int = 0,
                 world: int = 1, group=None, send_fraction: float = 0.25, exchange_interval: int = 1, comm=None):
        self.ops, self.rank, self.world, self.group = ops, rank, world, group
        # the collectives: torch.distributed (backend "nccl" = RCCL) unless a stand-in with the same three
        # functions is injected (tests/test_gpu_two_ranks.py runs two ranks as threads on one GPU)
        self.comm = dist if comm is None else comm
        self.device = device
        self.capacity = int(capacity)
        f64 = dict(dtype=torch.float64, device=device)
        self.x = torch.empty(self.capacity, **f64)
        self.y = torch.empty(self.capacity, **f64)
        self.z = torch.empty(self.capacity, **f64)
        self.cell = torch.full((self.capacity,), L.CELL_LOST, dtype=torch.int32, device=device)
        self.gid = torch.zeros(self.capacity, dtype=torch.int64, device=device)
        self.n = 0
        self.cell_lo = np.asarray(cell_lo, dtype=np.int32)
        assert self.cell_lo.size == world + 1
        self.cell_lo_dev = torch.from_numpy(self.cell_lo.copy()).to(device)
        self.send_capacity = max(1024, int(self.capacity * send_fraction))
        self.sendbuf = torch.empty(self.send_capacity * L.HANDOFF_DOUBLES, **f64)
        self.recvbuf = torch.empty(self.send_capacity * L.HANDOFF_DOUBLES, **f64)
        # per-destination leaver counts (the split kernels write up to kMaxRanks = 16 of them) and nStay sit in ONE device
        # tensor, so that a hand-off's bookkeeping is one all-gather into one table and ONE copy into one pinned host
        # buffer: [world x (16 + 1)] counts | nStay rows, then this rank's cell_lo (world + 1)
        if world > 16:
            raise ValueError("ShardedCloud: at most 16 ranks (the split kernels' count vector)")
        self.meta_dev = torch.zeros(17, dtype=torch.int64, device=device)
        self.counts_dev = self.meta_dev[:16]
        self.nstay_dev = self.meta_dev[16:17]
        self._table_dev = torch.zeros(world * 17 + world + 1, dtype=torch.int64, device=device)
        self._table_host = torch.zeros(world * 17 + world + 1, dtype=torch.int64,
                                       pin_memory=(device.type == "cuda"))
        # overlapped hand-off: after the split the step loop runs on for ``overlap_steps`` cycles while counts and
        # payload travel on a side stream; the arrivals then catch up on the cycles they missed (see step())
        self.overlap_steps = 0
        self.timing_on = False       # set when the caller brackets step launches with events itself (bench.py)
        self._pending = None
        self._sort_due = False
        self._step_args = None
        self._side = torch.cuda.Stream(device=device) if device.type == "cuda" else None
        self.weights_dev = None      # per-cell particle counts x cost (allocated by the first rebalance)
        # load balancing by MEASURED step time (see rebalance()): ms per particle-step of this rank's region
        self.balance_by_time = False
        self.cost_per_particle = None
        self.kernel_ms = 0.0         # step-kernel device time drained by the balancer (ms) ...
        self.kernel_launches = 0     # ... and the launches it covers (bench.py adds them to its own read)
        self.exchange_interval = max(0, int(exchange_interval))   # 0 = only hand off inside rebalance()
        self.step_index = 0
        self.particle_steps = 0      # cumulative particles x steps stepped by this rank
        self.handed_off = 0          # cumulative particles sent away by this rank
        self.exchanges = 0
        self.rebalances = 0
        self.grown = 0               # times the arrays had to be enlarged for arrivals
        self.handoff_host_ms = 0.0   # host wall time spent in the re-cut / split / exchange calls (incl. their one sync)
        self._host_work_ms = 0.5     # running mean of the host's own work per hand-off, without the wait (see _overlap)
        self.handoff_wait_ms = 0.0   # ... of which: blocked in that one sync (the device catching up with the queued steps + the counts' all-gather)
        self.profile_comm = False    # keep timing-enabled (start, end) device events around every hand-off's collectives
        self._comm_events = []       # ... here (bench.py reads them); off: one plain event per hand-off, nothing kept
        self.send_grown = 0          # times the send buffer had to be enlarged and the split repeated (see _finish_exchange)
        self.rebalance_interval = 0  # 0 = never; else every that many steps (needs n_cells)
        self.sort_interval = 0       # 0 = never; else re-sort by cell every that many steps (coalescing)
        self.force_collectives = False   # run the hand-off path even with one rank (single-GPU smoke of the N>1 code)
        self.n_cells = int(self.cell_lo[-1])

    # -- filling
    def set_particles(self, x: torch.Tensor, y: torch.Tensor, z: torch.Tensor, cell: Optional[torch.Tensor],
                      gid: torch.Tensor):
        n = int(x.numel())
        if n > self.capacity:
            raise ValueError("shard capacity %d < %d particles" % (self.capacity, n))
        self.x[:n].copy_(x); self.y[:n].copy_(y); self.z[:n].copy_(z); self.gid[:n].copy_(gid)
        self.n = n
        if cell is None:
            self.ops.locate(self)
        else:
            self.cell[:n].copy_(cell)

    def set_velocity(self, U):
        """New cell velocities (transient solvers, src/advect.H:44-57).  A hand-off still in flight is completed
        FIRST: its arrivals replay the cycles they missed in one launch, and that replay must see the field those
        cycles were stepped with, not the new one."""
        self._finish_exchange()
        self.ops.set_velocity(U)

    def comm_ms(self) -> float:
        """Device time of the hand-off collectives so far (counts all-gather + payload all-to-all, side stream)."""
        tot = 0.0
        for a, b in self._comm_events:
            b.synchronize()
            tot += a.elapsed_time(b)
        return tot

    # -- the hot loop
    def step(self, dt: float, n_cycles: int = 1, D: float = 0.0, flags: int = 0):
        if self._pending is not None and self._step_args != (dt, D, flags):
            self._finish_exchange()                     # the catch-up replays the window with ONE set of arguments
        self._step_args = (dt, D, flags)
        dist_on = self.world > 1 or self.force_collectives
        for _ in range(n_cycles):
            if self._pending is not None and self.step_index - self._pending["step"] >= self._overlap():
                self._finish_exchange()
            self.ops.step(self, dt, D, self.step_index, 1, flags)
            self.step_index += 1
            self.particle_steps += self.n
            if self.sort_interval and self.step_index % self.sort_interval == 0:
                if self._pending is None:
                    self.sort()
                else:
                    self._sort_due = True               # never reorder while stale tail slots are in the range
            if dist_on:
                if self.rebalance_interval and self.step_index % self.rebalance_interval == 0:
                    self._finish_exchange()
                    self._recut()
                    self._begin_exchange()
                elif self.exchange_interval and self.step_index % self.exchange_interval == 0:
                    self._finish_exchange()
                    self._begin_exchange()
        if self.overlap_steps == 0:
            self._finish_exchange()

    def _overlap(self) -> int:
        """Steps the loop runs on between a split and its exchange.  ``overlap_steps >= 0``: that many.  ``overlap_steps < 0``
        (auto): enough queued steps to cover the host's own work per hand-off -- everything it does after the one wait:
        reading the table, enqueueing the all-to-all, the unpack, the catch-up launch -- measured as a running mean, against
        this rank's step time (measured when the balancer times the launches, else the nominal rate): ceil(host / step) + 1,
        at least 2, at most half the hand-off interval.  Ranks may choose differently: every rank issues the same sequence
        of collectives whatever its depth, and the catch-up replays what each rank itself missed."""
        if self.overlap_steps >= 0:
            return self.overlap_steps
        step_ms = (self.cost_per_particle or 1.0) * COST_UNIT_MS * max(self.n, 1)
        interval = self.rebalance_interval or self.exchange_interval or 16
        want = int(np.ceil(self._host_work_ms / max(step_ms, 1e-3))) + 1
        return int(min(max(want, 2), max(1, interval // 2)))

    def _grow(self, needed: int, n_keep: int):
        """More arrivals than slack: move the shard into larger arrays (HBM is plentiful; on the compute stream,
        so it is ordered after the steps in flight)."""
        cap = max(int(needed), int(self.capacity * 1.5)) + 4096
        for name in ("x", "y", "z", "cell", "gid"):
            old = getattr(self, name)
            new = torch.empty(cap, dtype=old.dtype, device=self.device)
            new[:n_keep].copy_(old[:n_keep])
            setattr(self, name, new)
        if self.send_capacity >= self.capacity:                       # "a send buffer as large as the shard" stays so
            self.send_capacity = cap
            self.sendbuf = torch.empty(cap * L.HANDOFF_DOUBLES, dtype=torch.float64, device=self.device)
        self.capacity = cap
        self.grown += 1

    def flush(self):
        """Completes a hand-off still in flight (arrivals appended and caught up)."""
        self._finish_exchange()

    def exchange(self):
        """Hand particles that left this rank's cell range to their owners (all-to-all-v), synchronously."""
        if self.world == 1 and not self.force_collectives:
            return
        self._finish_exchange()
        self._begin_exchange()
        self._finish_exchange()

    def _begin_exchange(self):
        """Split the shard on the compute stream: leavers into the send buffer, stayers compacted into
        [0, nStay), the stale tail marked inactive.  Counts and nStay stay in device memory for now."""
        t_host = time.perf_counter()
        self.ops.pack(self)
        ev = None
        if self._side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        self._pending = {"step": self.step_index, "event": ev}
        self.handoff_host_ms += (time.perf_counter() - t_host) * 1e3

    def _finish_exchange(self):
        """Counts all-gather + payload all-to-all-v on the side stream (the compute stream keeps running the
        steps queued since the split), then append the arrivals and let them catch up on those steps.

        One host synchronisation, on the side stream only: the per-destination counts of every rank are
        all-gathered on the device (world x (world+1) int64) and copied to the host once, which gives this
        rank both its send sizes (its row) and its receive sizes (its column).

        Send-buffer overflow cannot lose particles or hang the job: a split whose leavers do not fit reports
        ``nStay < 0`` and moves NOTHING (cpf_pack_leavers_dev).  Every rank reads that in the same all-gathered
        table, so all of them take the same branch: the overflowing ranks enlarge their send buffers and split again
        (on the compute stream, i.e. at the current step: their leavers then need no catch-up), everybody repeats
        the all-gather, and only then does the all-to-all run."""
        p = self._pending
        if p is None:
            return
        t_host = time.perf_counter()
        wait0 = self.handoff_wait_ms
        self._pending = None
        W = self.world
        D = L.HANDOFF_DOUBLES
        cuda = self._side is not None
        compute = torch.cuda.current_stream(self.device) if cuda else None
        side = torch.cuda.stream(self._side) if cuda else contextlib.nullcontext()
        repacked = set()                # ranks whose split was repeated at the current step
        with side:
            if cuda:
                self._side.wait_event(p["event"])
                ev0 = torch.cuda.Event(enable_timing=self.profile_comm); ev0.record(self._side)
            attempts = 0
            while True:
                # ONE all-gather into the device table, the rank's cuts behind it, ONE copy into the pinned host buffer
                gathered = self._table_dev[: W * 17]
                if hasattr(self.comm, "all_gather_into_tensor"):
                    self.comm.all_gather_into_tensor(gathered, self.meta_dev, group=self.group)
                else:
                    self.comm.all_gather(list(gathered.view(W, 17).unbind(0)), self.meta_dev, group=self.group)
                self._table_dev[W * 17:].copy_(self.cell_lo_dev)
                self._table_host.copy_(self._table_dev, non_blocking=True)
                if cuda:
                    got = torch.cuda.Event(); got.record(self._side)
                    t_wait = time.perf_counter()
                    got.synchronize()                                    # the hand-off's one host wait (side stream only)
                    self.handoff_wait_ms += (time.perf_counter() - t_wait) * 1e3
                host = self._table_host.numpy()
                table = host[: W * 17].reshape(W, 17)
                over = [r for r in range(W) if table[r, 16] < 0]
                if not over:
                    break
                attempts += 1
                if attempts > 8:
                    raise RuntimeError("hand-off: rank(s) %s still overflow after %d enlargements of their send buffers"
                                       % (over, attempts - 1))             # same table, same exception on every rank
                if self.rank in over:
                    # The repeated split runs at the CURRENT step: the leavers have kept accumulating since the aborted one
                    # (this rank moved nothing in between), so size for that -- and if it still does not fit, the loop
                    # simply grows again: every rank sees the same table and takes the same branch.
                    need = int(table[self.rank, :W].sum())
                    missed = max(0, self.step_index - p["step"])
                    grow = need * (1 + missed) if attempts == 1 else max(need, self.send_capacity) * 2
                    self.send_capacity = min(max(grow + grow // 8 + 1024, self.send_capacity), self.capacity)
                    with torch.cuda.stream(compute) if cuda else contextlib.nullcontext():
                        self.sendbuf = torch.empty(self.send_capacity * D, dtype=torch.float64, device=self.device)
                    self.send_grown += 1
                    self.ops.pack(self)                                  # compute stream: after the steps queued so far
                    if cuda:
                        self.sendbuf.record_stream(self._side)           # allocated on the compute stream, read by the side stream
                        ev = torch.cuda.Event(); ev.record(compute)
                        self._side.wait_event(ev)
                repacked |= set(over)
            self.cell_lo = host[W * 17:].astype(np.int32)
            send_counts = [int(v) for v in table[self.rank, :W]]
            recv_counts = [int(v) for v in table[:, self.rank]]
            n_stay = int(table[self.rank, 16])
            n_send, n_recv = sum(send_counts), sum(recv_counts)
            assert n_send <= self.send_capacity
            if n_recv * D > self.recvbuf.numel():
                self.recvbuf = torch.empty(n_recv * D, dtype=torch.float64, device=self.device)
                if cuda:
                    self.recvbuf.record_stream(compute)                  # unpack reads it there
            self.comm.all_to_all_single(self.recvbuf[: n_recv * D], self.sendbuf[: n_send * D],
                                   [c * D for c in recv_counts], [c * D for c in send_counts], group=self.group)
            if cuda:
                done = torch.cuda.Event(enable_timing=self.profile_comm)
                done.record(self._side)
                if self.profile_comm:
                    self._comm_events.append((ev0, done))
        if cuda:
            compute.wait_event(done)   # also orders the next pack after this all-to-all
        missed = self.step_index - p["step"]
        if self.rank not in repacked:
            self.particle_steps -= (self.n - n_stay) * missed         # the inactive tail was not real work
        if n_stay + n_recv > self.capacity:
            self._grow(n_stay + n_recv, n_stay)
        self.ops.unpack(self, n_stay, self.recvbuf, n_recv)
        if missed and n_recv:
            # arrivals sit in source-rank order; those from a rank that split again at the current step are current,
            # the others replay the cycles they missed -- one launch per run of consecutive sources
            dt, Dc, flags = self._step_args
            first = n_stay
            run_first, run_count = first, 0
            for src in range(W):
                k = recv_counts[src]
                if src in repacked:
                    if run_count:
                        self.ops.step_slice(self, run_first, run_count, dt, Dc, p["step"], missed, flags)
                        self.particle_steps += run_count * missed
                    run_first, run_count = first + k, 0
                else:
                    run_count += k
                first += k
            if run_count:
                self.ops.step_slice(self, run_first, run_count, dt, Dc, p["step"], missed, flags)
                self.particle_steps += run_count * missed
        self.n = n_stay + n_recv
        self.handed_off += n_send
        self.exchanges += 1
        if self._sort_due:
            self._sort_due = False
            self.sort()
        total = (time.perf_counter() - t_host) * 1e3
        self.handoff_host_ms += total
        self._host_work_ms = 0.5 * self._host_work_ms + 0.5 * max(0.0, total - (self.handoff_wait_ms - wait0))

    def rebalance(self, n_cells: Optional[int] = None):
        """Re-cut the ranges (see ``_recut``) and hand particles to their new owners, synchronously."""
        if self.world == 1 and not self.force_collectives:
            return
        self._finish_exchange()
        self._recut(n_cells)
        self._begin_exchange()
        self._finish_exchange()                # also refreshes the host copy self.cell_lo

    def _recut(self, n_cells: Optional[int] = None):
        """Re-cut the cell ranges so that every rank owns the same number of particles (or the same
        measured cost), then hand particles to their new owners.  Per-cell histogram (HIP kernel) ->
        all-reduce -> prefix sums and cut search in one HIP kernel (no host round trip; the rule of
        ``slab_cell_ranges``) -> exchange.

        Legal at any time because the mesh is replicated.  For a cloud that drifts with the flow the
        equal-count cuts drift with it, so re-cutting hands over far fewer particles than keeping the
        ranges fixed would (and nothing piles up on the outlet rank)."""
        t_host = time.perf_counter()
        n_cells = self.n_cells if n_cells is None else int(n_cells)
        if self.weights_dev is None or self.weights_dev.numel() != n_cells:
            self.weights_dev = torch.zeros(n_cells, dtype=torch.float64, device=self.device)
        # equal-COST cuts when balancing by time: every rank scales its counts by its measured ms per
        # particle-step (hops per step differ across the mesh: fine cells cost more), so the all-reduced
        # histogram is a cost density; scale 1 gives equal-count cuts
        self.ops.histogram(self, self._measured_cost() if self.balance_by_time else 1.0)
        self.comm.all_reduce(self.weights_dev, group=self.group)
        self.ops.cell_ranges(self)
        self.rebalances += 1
        self.handoff_host_ms += (time.perf_counter() - t_host) * 1e3

    def enable_time_balancing(self, on: bool = True):
        """Re-cut by measured cost instead of by particle count (must be set the same on every rank)."""
        self.balance_by_time = bool(on)
        if on:
            self.ops.enable_timing(self)

    def _measured_cost(self) -> float:
        """This rank's cost per particle-step in units of COST_UNIT_MS (about 1 at the 1e7-particle bench rate;
        the unit only has to be the same on every rank), from the HIP-event times of its own step launches.
        Never stalls the launch queue except for the very first measurement; smoothed 50/50 with the previous
        value.  Before any step has run every rank returns 1 (plain equal-count cuts)."""
        first = self.cost_per_particle is None
        launches, ms = self.ops.step_time(self, wait=first)
        self.kernel_ms += ms; self.kernel_launches += launches
        if launches > 0 and self.n > 0:
            # clamped: one rank's bad measurement must not pull most of the cloud onto another rank
            cost = min(4.0, max(0.25, ms / launches / self.n / COST_UNIT_MS))
            self.cost_per_particle = cost if first else 0.5 * (self.cost_per_particle + cost)
        return 1.0 if self.cost_per_particle is None else self.cost_per_particle

    def sort(self):
        self.ops.sort(self)

    # -- inspection
    def gather_to_numpy(self):
        """(gid, x, y, z, cell) of this shard on the host."""
        self._finish_exchange()
        n = self.n
        return (self.gid[:n].cpu().numpy(), self.x[:n].cpu().numpy(), self.y[:n].cpu().numpy(),
                self.z[:n].cpu().numpy(), self.cell[:n].cpu().numpy())

    def global_count(self) -> int:
        self._finish_exchange()
        t = torch.tensor([self.n], dtype=torch.int64, device=self.device)
        if self.world > 1:
            self.comm.all_reduce(t, group=self.group)
        return int(t.item())
