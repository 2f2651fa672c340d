"""GPU: the sharded cloud with SEVERAL ranks on ONE MI355X, entirely through the C-ABI.

The development loop has a single GPU per box and RCCL refuses two ranks on one device, so the N > 1 path would
otherwise meet real leavers for the first time in the driver's scaling run.  Here the ranks run as threads of this
process, each with its own context, streams and shard (cpf_shard_*); the collectives are the library's IN-PROCESS
communicator (cpf_comm_create with a CPF_COMM_INPROCESS token: device-to-device copies and a barrier,
csrc/cpf_comm.cpp).  Everything else is what an 8-GPU run executes: the HIP split / histogram / cut / unpack kernels with
real leavers, the overlapped hand-off with its catch-up launch, growth of a shard, balancing by measured step time, all
strung together by csrc/cpf_shard_core.h.  The answer is the single-process CPU statement, particle by particle.
"""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
os.environ.setdefault("CPF_COMM_TIMEOUT", "180")      # a rank that died must not leave the others waiting at the barrier for long


def _run_two_ranks(pitz, oracle_libs, *, n_total, steps, rebalance, exchange, overlap, balance_by_time, capacity,
                   dt=1e-4, world=2, send_fraction=1.0, step_flags=0):
    import torch
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.parallel import Communicator, ShardedCloud, slab_cell_ranges, unique_id, x_slab_renumbering
    pz = pitz["pz"]
    c0, _ = pitz["mesh"].cell_centres_volumes()
    mesh = pitz["mesh"].renumber_cells(x_slab_renumbering(c0))
    centres, vols = mesh.cell_centres_volumes()
    U = pz.analytic_step_u(mesh, centres)
    cw = oracle_libs.CellWalk(); t = cw.build(mesh)
    xyz = pz.uniform_points(321, n_total, *pz.DOMAIN_BOX)
    x, y, z = (xyz[:, k].copy() for k in range(3))
    cell = cw.locate_initial(x, y, z, t, nthreads=cw.max_threads)
    token = unique_id(L.COMM_INPROCESS)
    cell_lo = slab_cell_ranges(vols, world)
    dev = torch.device("cuda", 0)
    out, errors = [None] * world, []

    def rank_main(rank):
        try:
            ctx = Context(0)                                      # its own stream
            ctx.set_mesh(mesh); ctx.set_velocity(U)
            comm = Communicator(token, rank, world, 0)
            mine = np.arange(rank, n_total, world)                # an arbitrary share: the first hand-off sorts it out
            cloud = ShardedCloud(ctx, cell_lo, capacity, comm, send_fraction=send_fraction, exchange_interval=exchange)
            cloud.rebalance_interval = rebalance
            cloud.overlap_steps = overlap
            cloud.sort_interval = 5
            if balance_by_time:
                cloud.enable_time_balancing()
            tx, ty, tz = (torch.from_numpy(a[mine].copy()).to(dev) for a in (x, y, z))
            tg = torch.from_numpy(mine.astype(np.int64)).to(dev)
            torch.cuda.synchronize()
            cloud.set_particles(tx, ty, tz, None, tg)
            cloud.exchange()
            cloud.step(dt, steps, flags=step_flags)
            cloud.flush()
            g, gx, gy, gz, gc = cloud.gather_to_numpy()
            total = cloud.global_count()
            whole = cloud.gather(0)                               # collective: the cloud in particle-id order on rank 0
            out[rank] = dict(g=g, x=gx, y=gy, z=gz, c=gc, handed=cloud.handed_off, grown=cloud.grown, send_grown=cloud.send_grown,
                             rebalances=cloud.rebalances, cell_lo=cloud.cell_lo.copy(), n=cloud.n, total=total, whole=whole)
            cloud.close(); comm.close(); ctx.close()
        except BaseException as e:                                  # noqa: BLE001 -- reported by the main thread
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=900)
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
        assert o["total"] == n_total
    assert seen.all()
    xyzw, wc, _ = out[0]["whole"]
    assert np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) and np.array_equal(xyzw[:, 2], z) and np.array_equal(wc, cell)
    assert all(o["whole"][0] is None for o in out[1:])
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
    """`overlap_steps = -1` (bench.py's default since round 4): every rank derives how many steps it would queue between a split
    and its exchange from its own measured host work per hand-off and its own step time; the ranks agree on the maximum through
    the counts table (csrc/cpf_shard_core.h, depthNext), and every particle still equals the one-process run bit for bit."""
    out, cell = _run_two_ranks(pitz, oracle_libs, n_total=400_000, steps=32, rebalance=8, exchange=0, overlap=-1,
                               balance_by_time=True, capacity=400_000 + 64)
    assert all(o["rebalances"] == 4 for o in out) and sum(o["handed"] for o in out) > 10_000
    assert sum(o["n"] for o in out) == 400_000


def test_two_ranks_fused_cycles_between_the_triggers(pitz, oracle_libs):
    """CPF_STEP_FUSE_CYCLES through cpf_shard_step (what the parallel fragments pass between two frames): the cycles up to the next
    sort (every 5), re-cut (every 8) or completion of the hand-off in flight (derived depth) share a launch; the particles still
    equal the one-process run bit for bit."""
    from cudaparticlesfoam_amd import _lib as L
    out, cell = _run_two_ranks(pitz, oracle_libs, n_total=400_000, steps=32, rebalance=8, exchange=0, overlap=-1,
                               balance_by_time=True, capacity=400_000 + 64, step_flags=L.STEP_FUSE_CYCLES)
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


def test_two_ranks_send_buffer_overflow_with_derived_depth(pitz, oracle_libs):
    """A send buffer of 1 % of the shard with `overlap_steps = -1`: the first hand-offs overflow, the split aborts on the device,
    every rank reads that in the all-gathered table, the overflowing ranks enlarge their buffers and split again at the current
    step -- the same step on every rank, because the depth is agreed -- and no particle is lost, duplicated or left behind."""
    out, cell = _run_two_ranks(pitz, oracle_libs, n_total=300_000, steps=20, rebalance=0, exchange=2, overlap=-1,
                               balance_by_time=False, capacity=300_000 + 64, send_fraction=0.01)
    assert sum(o["send_grown"] for o in out) >= 2 and sum(o["n"] for o in out) == 300_000
