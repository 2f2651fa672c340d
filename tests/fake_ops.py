"""Test double for the device operations of a shard (CPU, numpy + the cell-walk oracle).

Lets the world_size-2 gloo tests exercise the HOST logic of cudaparticlesfoam_amd/parallel.py
(ownership ranges, counts exchange, variable-size all-to-all, append) without a GPU.  This lives
under tests/ on purpose: the product has no CPU path.
"""
import numpy as np
import torch

from cudaparticlesfoam_amd import _lib as L


class FakeOps:
    def __init__(self, cellwalk, tables, U, fake_ms_per_particle=2.0e-8):
        self.cw, self.t, self.U = cellwalk, tables, np.ascontiguousarray(U, dtype=np.float64)
        self.fake_ms_per_particle = fake_ms_per_particle      # pretend step cost of THIS rank (load-balancer tests)
        self._launches, self._ms = 0, 0.0

    def enable_timing(self, s):
        pass

    def set_velocity(self, U):
        self.U = np.ascontiguousarray(U, dtype=np.float64)

    def step_time(self, s, wait):
        out = (self._launches, self._ms)
        self._launches, self._ms = 0, 0.0
        return out

    def step(self, s, dt, D, step0, n_cycles, flags):
        n = s.n
        self._launches += 1; self._ms += n * self.fake_ms_per_particle
        x, y, z = (a[:n].numpy().copy() for a in (s.x, s.y, s.z))
        c = s.cell[:n].numpy().copy()
        g = s.gid[:n].numpy().copy()
        self.cw.step(x, y, z, c, dt, n_cycles, self.t, self.U, nthreads=1, D=D, gid=g, step0=step0, seed=0)
        for dst, src in ((s.x, x), (s.y, y), (s.z, z), (s.cell, c)):
            dst[:n] = torch.from_numpy(src)

    def step_slice(self, s, first, count, dt, D, step0, n_cycles, flags):
        sl = slice(first, first + count)
        x, y, z = (a[sl].numpy().copy() for a in (s.x, s.y, s.z))
        c = s.cell[sl].numpy().copy()
        g = s.gid[sl].numpy().copy()
        self.cw.step(x, y, z, c, dt, n_cycles, self.t, self.U, nthreads=1, D=D, gid=g, step0=step0, seed=0)
        for dst, src in ((s.x, x), (s.y, y), (s.z, z), (s.cell, c)):
            dst[sl] = torch.from_numpy(src)

    def pack(self, s):
        n = s.n
        cell = s.cell[:n].numpy()
        owner = np.searchsorted(s.cell_lo_dev.numpy()[1:], cell, side="right")   # the device copy is the authority
        dest = np.where((cell < 0) | (owner == s.rank), -1, owner)
        stay = np.nonzero(dest < 0)[0]
        recs = []
        counts = np.zeros(16, np.int64)
        for r in range(s.world):
            idx = np.nonzero(dest == r)[0]
            counts[r] = idx.size
            recs.append(np.stack([s.x[:n].numpy()[idx], s.y[:n].numpy()[idx], s.z[:n].numpy()[idx],
                                  cell[idx].astype(np.float64), s.gid[:n].numpy()[idx].astype(np.float64)], 1))
        s.counts_dev[:] = torch.from_numpy(counts)
        if int(counts.sum()) > s.send_capacity:                 # like the HIP split: aborted, nothing moved
            s.nstay_dev[0] = -1
            return
        flat = np.concatenate(recs).reshape(-1)
        s.sendbuf[: flat.size] = torch.from_numpy(flat)
        for a in (s.x, s.y, s.z, s.cell, s.gid):
            a[: stay.size] = a[:n][torch.from_numpy(stay)].clone()
        s.cell[stay.size:n] = L.CELL_LOST                       # like the HIP split: stale tail slots are inactive
        s.x[stay.size:n] = float("nan")                          # ... and must never be read again
        s.counts_dev[:] = torch.from_numpy(counts)
        s.nstay_dev[0] = stay.size

    def unpack(self, s, n_stay, recvbuf, n_recv):
        if n_recv == 0:
            return
        rec = recvbuf[: n_recv * L.HANDOFF_DOUBLES].reshape(n_recv, L.HANDOFF_DOUBLES)
        s.x[n_stay:n_stay + n_recv] = rec[:, 0]; s.y[n_stay:n_stay + n_recv] = rec[:, 1]
        s.z[n_stay:n_stay + n_recv] = rec[:, 2]
        s.cell[n_stay:n_stay + n_recv] = rec[:, 3].to(torch.int32)
        s.gid[n_stay:n_stay + n_recv] = rec[:, 4].to(torch.int64)

    def histogram(self, s, scale):
        c = s.cell[: s.n].numpy()
        s.weights_dev[:] = torch.from_numpy(np.bincount(c[c >= 0], minlength=s.weights_dev.numel()) * float(scale))

    def cell_ranges(self, s):
        from cudaparticlesfoam_amd.parallel import device_cell_ranges
        s.cell_lo_dev.copy_(device_cell_ranges(s.weights_dev, s.world))

    def sort(self, s):
        n = s.n
        order = torch.argsort(s.cell[:n].to(torch.int64) & 0xFFFFFFFF, stable=True)
        for a in (s.x, s.y, s.z, s.cell, s.gid):
            a[:n] = a[:n][order].clone()

    def locate(self, s):
        n = s.n
        c = self.cw.locate_initial(s.x[:n].numpy().copy(), s.y[:n].numpy().copy(), s.z[:n].numpy().copy(), self.t)
        s.cell[:n] = torch.from_numpy(c)
