"""GPU: the sharded cloud with TWO ranks on ONE MI355X.

The development loop has a single GPU per box and RCCL refuses two ranks on one device, so the N > 1 host path
would otherwise meet real leavers for the first time in the driver's scaling run.  Here both ranks run as
threads of this process, each with its own context, streams and ShardedCloud; only the three collectives are
replaced by an in-process stand-in (ThreadComm, below).  Everything else is the product: the HIP split /
histogram / cut / unpack kernels with real leavers, the overlapped hand-off with its catch-up launch, growth of
a shard, balancing by measured step time.  The answer is the single-process CPU statement, particle by particle.
"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class ThreadComm:
    """all_gather / all_reduce / all_to_all_single between threads of one process (test double for the
    torch.distributed calls of cudaparticlesfoam_amd/parallel.py; device tensors, lock-step like a collective)."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.tls = threading.local()

    def bind(self, rank):
        self.tls.rank = rank

    def _sync(self):
        import torch
        torch.cuda.current_stream().synchronize()

    def all_gather(self, rows, t, group=None):
        r = self.tls.rank
        self._sync()
        self.slots[r] = t.clone()
        self._sync(); self.barrier.wait()
        for k in range(self.world):
            rows[k].copy_(self.slots[k])
        self._sync(); self.barrier.wait()

    def all_reduce(self, t, group=None):
        r = self.tls.rank
        self._sync()
        self.slots[r] = t.clone()
        self._sync(); self.barrier.wait()
        total = self.slots[0].clone()
        for k in range(1, self.world):
            total += self.slots[k]
        self._sync(); self.barrier.wait()
        t.copy_(total)
        self._sync(); self.barrier.wait()

    def all_to_all_single(self, out, inp, out_splits, in_splits, group=None):
        r = self.tls.rank
        self._sync()
        self.slots[r] = (inp.clone(), list(in_splits))
        self._sync(); self.barrier.wait()
        off_out = 0
        for k in range(self.world):
            src, splits = self.slots[k]
            a = sum(splits[:r]); n = splits[r]
            assert n == out_splits[k]
            out[off_out:off_out + n].copy_(src[a:a + n])
            off_out += n
        self._sync(); self.barrier.wait()


def _run_two_ranks(pitz, oracle_libs, *, n_total, steps, rebalance, exchange, overlap, balance_by_time, capacity,
                   dt=1e-4, world=2):
    import torch
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.parallel import HipOps, ShardedCloud, slab_cell_ranges, x_slab_renumbering
    pz = pitz["pz"]
    c0, _ = pitz["mesh"].cell_centres_volumes()
    mesh = pitz["mesh"].renumber_cells(x_slab_renumbering(c0))
    centres, vols = mesh.cell_centres_volumes()
    U = pz.analytic_step_u(mesh, centres)
    cw = oracle_libs.CellWalk(); t = cw.build(mesh)
    xyz = pz.uniform_points(321, n_total, *pz.DOMAIN_BOX)
    x, y, z = (xyz[:, k].copy() for k in range(3))
    cell = cw.locate_initial(x, y, z, t, nthreads=cw.max_threads)
    comm = ThreadComm(world)
    cell_lo = slab_cell_ranges(vols, world)
    dev = torch.device("cuda", 0)
    out, errors = [None] * world, []

    def rank_main(rank):
        try:
            comm.bind(rank)
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                ctx = Context(0)
                ctx.set_stream(stream.cuda_stream)
                ctx.set_mesh(mesh); ctx.set_velocity(U)
                mine = np.arange(rank, n_total, world)            # an arbitrary half: the first hand-off sorts it out
                cloud = ShardedCloud(HipOps(ctx), cell_lo, capacity, dev, rank, world, send_fraction=1.0,
                                     exchange_interval=exchange, comm=comm)
                cloud.rebalance_interval = rebalance
                cloud.overlap_steps = overlap
                cloud.sort_interval = 5
                if balance_by_time:
                    cloud.enable_time_balancing()
                cloud.set_particles(*(torch.from_numpy(a[mine].copy()).to(dev) for a in (x, y, z)), None,
                                    torch.from_numpy(mine.astype(np.int64)).to(dev))
                cloud.exchange()
                cloud.step(dt, steps)
                cloud.flush()
                g, gx, gy, gz, gc = cloud.gather_to_numpy()
                out[rank] = dict(g=g, x=gx, y=gy, z=gz, c=gc, handed=cloud.handed_off, grown=cloud.grown,
                                 rebalances=cloud.rebalances, cell_lo=cloud.cell_lo.copy(), n=cloud.n)
                torch.cuda.current_stream().synchronize()
                ctx.close()
        except BaseException as e:                                  # noqa: BLE001 -- reported by the main thread
            errors.append((rank, repr(e)))
            comm.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not errors, errors
    assert all(o is not None for o in out)
    cw.step(x, y, z, cell, dt, steps, t, U, nthreads=cw.max_threads)
    seen = np.zeros(n_total, bool)
    for o in out:
        g = o["g"]
        assert not seen[g].any()
        seen[g] = True
        assert np.array_equal(o["c"], cell[g])
        assert np.array_equal(o["x"], x[g]) and np.array_equal(o["y"], y[g]) and np.array_equal(o["z"], z[g])
    assert seen.all()
    return out, cell


def test_two_ranks_drifting_cuts_overlapped_handoff(pitz, oracle_libs):
    """Re-cut + hand-off every 4 steps, 2 steps overlapped, ranges balanced by measured step time."""
    out, cell = _run_two_ranks(pitz, oracle_libs, n_total=400_000, steps=24, rebalance=4, exchange=0, overlap=2,
                               balance_by_time=True, capacity=400_000 + 64)
    assert all(o["rebalances"] == 6 for o in out) and sum(o["handed"] for o in out) > 10_000
    assert np.array_equal(out[0]["cell_lo"], out[1]["cell_lo"])      # both ranks cut at the same cells
    lo = out[0]["cell_lo"]
    for r, o in enumerate(out):                                       # the last re-cut came after the last step
        assert ((o["c"] >= lo[r]) & (o["c"] < lo[r + 1]) | (o["c"] < 0)).all()
    assert abs(out[0]["n"] - out[1]["n"]) < 0.25 * 400_000            # equal cost, so roughly equal counts here


def test_two_ranks_at_2e7_particles(pitz, oracle_libs):
    """The two-rank path at the bench's per-GPU size (2 x 1e7): re-cut + hand-off every 6 steps, 2 overlapped,
    balanced by measured time -- still bit-identical to one process."""
    out, cell = _run_two_ranks(pitz, oracle_libs, n_total=20_000_000, steps=12, rebalance=6, exchange=0, overlap=2,
                               balance_by_time=True, capacity=20_000_000 + 64)
    assert all(o["rebalances"] == 2 for o in out) and sum(o["handed"] for o in out) > 100_000
    assert sum(o["n"] for o in out) == 20_000_000


def test_two_ranks_fixed_ranges_growing_shard(pitz, oracle_libs):
    """Fixed x-slabs, hand-off every 2 steps with 1 overlapped: the flow piles the cloud up on the downstream
    rank, whose arrays (capacity barely above the initial half) must grow on the device."""
    out, cell = _run_two_ranks(pitz, oracle_libs, n_total=300_000, steps=40, rebalance=0, exchange=2, overlap=1,
                               balance_by_time=False, capacity=160_000, dt=4e-4)
    assert sum(o["grown"] for o in out) >= 1 and out[1]["n"] > 160_000


def test_two_ranks_overlap_depth_derived_per_rank(pitz, oracle_libs):
    """`overlap_steps = -1` (bench.py's default since round 4): every rank derives how many steps it queues between a split and
    its exchange from its own measured host work per hand-off and its own step time (parallel.py, _overlap) -- the ranks may
    choose differently, and every particle still equals the one-process run bit for bit."""
    out, cell = _run_two_ranks(pitz, oracle_libs, n_total=400_000, steps=32, rebalance=8, exchange=0, overlap=-1,
                               balance_by_time=True, capacity=400_000 + 64)
    assert all(o["rebalances"] == 4 for o in out) and sum(o["handed"] for o in out) > 10_000
    assert sum(o["n"] for o in out) == 400_000


def test_four_ranks_as_threads_on_one_gpu(pitz, oracle_libs):
    """Four ranks (four contexts, streams and shards on ONE GPU, the collectives through the in-process stand-in): a four-way
    all-to-all with real leavers out of the HIP split kernels, re-cut every 6 steps by measured time, overlap depth derived per
    rank -- bit-identical to one process, every rank ending inside its own cut."""
    out, cell = _run_two_ranks(pitz, oracle_libs, n_total=600_000, steps=24, rebalance=6, exchange=0, overlap=-1,
                               balance_by_time=True, capacity=600_000 + 64, world=4)
    assert len(out) == 4 and all(o["rebalances"] == 4 for o in out) and sum(o["handed"] for o in out) > 10_000
    assert sum(o["n"] for o in out) == 600_000 and min(o["n"] for o in out) > 50_000
    lo = out[0]["cell_lo"]
    assert all(np.array_equal(o["cell_lo"], lo) for o in out)
    for r, o in enumerate(out):
        assert ((o["c"] >= lo[r]) & (o["c"] < lo[r + 1]) | (o["c"] < 0)).all()
