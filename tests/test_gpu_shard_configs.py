"""GPU: BASELINE.json's two multi-GPU configurations in their own shape and size, through the C path (cpf_comm_* / cpf_shard_*),
with the EIGHT ranks played by eight threads on the one GPU of this box (in-process communicator: RCCL refuses several ranks per
device).  No 8-GPU node has been available to any round; what these runs cannot show is xGMI bandwidth -- everything else an
8-GPU run executes (seeding split by rank, the re-cuts by measured time, the overlapped eight-way all-to-all-v with real leavers,
the catch-up replays, per-rank U slices all-gathered between the shards, the gather of a frame) runs here, and must give the
single-context run's particles bit for bit.

* configs[3]: 1e8 particles on pitzDaily sharded over 8 ranks.
* configs[4]: pimple-like -- a 440 100-cell mesh (records beyond L2), 1e7 particles, U(t) re-uploaded every Eulerian step as
  per-rank slices (src/advect.H:59-84), 8 ranks.
"""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
os.environ.setdefault("CPF_COMM_TIMEOUT", "300")


def _run_ranks(world, body):
    out, errs = [None] * world, []

    def main(r):
        try:
            out[r] = body(r)
        except BaseException as e:          # noqa: BLE001
            import traceback
            errs.append((r, repr(e), traceback.format_exc()))
    th = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(1500) for t in th]
    assert not errs, errs
    return out


def test_config3_1e8_particles_over_eight_ranks(pitz):
    """cpf_shard_seed_box(1e8) -- every rank draws its eighth of the single-GPU LCG stream, locates it, the ranges are cut to
    equal counts and every particle goes to its owner -- then 12 cycles with a re-cut by measured step time every 4, the overlap
    depth agreed between the ranks, and the gather of the whole cloud to rank 0: equal to cpf_seed_box + cpf_step on one context."""
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.parallel import Communicator, ShardedCloud, unique_id, x_slab_renumbering
    pz = pitz["pz"]
    c0, _ = pitz["mesh"].cell_centres_volumes()
    mesh = pitz["mesh"].renumber_cells(x_slab_renumbering(c0))
    centres, _ = mesh.cell_centres_volumes()
    U = pz.analytic_step_u(mesh, centres)
    n, world, k, dt = 100_000_000, 8, 12, 1e-4
    lo, hi = pz.DOMAIN_BOX
    with Context(0) as ctx:
        ctx.set_mesh(mesh); ctx.set_velocity(U)
        ctx.set_option("sort_interval", 0)
        ctx.seed_box(n, lo, hi, 1)
        n_out = ctx.locate_initial()
        ctx.step(dt, 0.0, k)
        want_xyzw, want_cell = ctx.get_particles()
    assert 0 < n_out < n // 4                                  # the box also covers the step: those particles are frozen, everywhere alike
    token = unique_id(L.COMM_INPROCESS)

    def body(r):
        with Context(0) as c:
            c.set_mesh(mesh); c.set_velocity(U)
            comm = Communicator(token, r, world, 0)
            cloud = ShardedCloud(c, None, n // world * 2, comm, send_fraction=0.5, exchange_interval=0)
            cloud.rebalance_interval = 4; cloud.overlap_steps = -1; cloud.sort_interval = 6
            cloud.enable_time_balancing()
            outside = cloud.seed_box(n, lo, hi, 1)
            cloud.step(dt, k)
            whole = cloud.gather(0)
            res = dict(outside=outside, n=cloud.n, handed=cloud.handed_off, rebalances=cloud.rebalances, lo=cloud.cell_lo.copy(),
                       whole=whole if r == 0 else None)
            cloud.close(); comm.close()
            return res
    out = _run_ranks(world, body)
    assert all(o["outside"] == n_out for o in out) and sum(o["n"] for o in out) == n
    assert all(o["rebalances"] == 1 + k // 4 for o in out) and sum(o["handed"] for o in out) > n // 2     # the seeding hand-off moves most of the cloud
    assert all(np.array_equal(o["lo"], out[0]["lo"]) for o in out) and min(o["n"] for o in out) > n // 40
    xyzw, cell, _ = out[0]["whole"]
    assert np.array_equal(cell, want_cell) and np.array_equal(xyzw, want_xyzw)


def test_config4_transient_field_on_a_440100_cell_mesh_over_eight_ranks(pitz):
    """The pimple-like configuration's shape: every Eulerian step each of the 8 ranks uploads ITS slice of the new field (an
    uneven split of the cells, like the pieces of a decomposed case) and the slices are all-gathered between the shards; 6
    cycles per step with hand-offs every 3 and two of them overlapped.  Equal to one context that is given the whole field."""
    import bench
    import torch
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.parallel import Communicator, ShardedCloud, unique_id, x_slab_renumbering
    pz = pitz["pz"]
    m0 = pz.pitzdaily_mesh(refine=6); c0, _ = m0.cell_centres_volumes()
    mesh = m0.renumber_cells(x_slab_renumbering(c0)); centres, _ = mesh.cell_centres_volumes()
    assert mesh.n_cells == 440_100
    base = pz.analytic_step_u(mesh, centres)
    fields = [base * (1.0 + 0.3 * np.sin(0.7 * e)) + np.array([0.0, 0.4 * np.cos(e), 0.0]) for e in range(3)]
    n, world, dt = 10_000_000, 8, 1e-4 / 6
    dev = torch.device("cuda", 0)
    p = lambda a: a.data_ptr()   # noqa: E731
    with Context(0) as ctx:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_mesh(mesh); ctx.set_velocity(base)
        x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 31, dev)
        start = [t.clone() for t in (x, y, z, c)]
        step = 0
        for U in fields:
            ctx.set_velocity(U)
            ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, dt, 0.0, step, 6, 0)
            step += 6
        torch.cuda.synchronize()
        want = [t.cpu().numpy() for t in (x, y, z, c)]
        ctx.use_own_stream()
    cut = [int(mesh.n_cells * (r / world) ** 1.3) for r in range(world)] + [mesh.n_cells]       # uneven pieces
    token = unique_id(L.COMM_INPROCESS)

    def body(r):
        with Context(0) as c2:
            c2.set_mesh(mesh); c2.set_velocity(base)
            comm = Communicator(token, r, world, 0)
            cloud = ShardedCloud(c2, None, n // world * 3, comm, send_fraction=1.0, exchange_interval=3)
            cloud.overlap_steps = 2; cloud.sort_interval = 5
            mine = slice(r * (n // world), (r + 1) * (n // world))           # an arbitrary eighth: the first hand-off sorts it out
            cloud.set_particles(start[0][mine], start[1][mine], start[2][mine], start[3][mine], None, first_gid=mine.start)
            cloud.rebalance()
            for U in fields:
                cloud.set_velocity_slice(U[cut[r]:cut[r + 1]])
                cloud.step(dt, 6)
            whole = cloud.gather(0)
            res = dict(n=cloud.n, handed=cloud.handed_off, exchanges=cloud.exchanges, whole=whole if r == 0 else None)
            cloud.close(); comm.close()
            return res
    out = _run_ranks(world, body)
    assert sum(o["n"] for o in out) == n and sum(o["handed"] for o in out) > n // 2 and all(o["exchanges"] >= 6 for o in out)
    xyzw, cell, _ = out[0]["whole"]
    assert np.array_equal(cell, want[3])
    assert np.array_equal(xyzw[:, 0], want[0]) and np.array_equal(xyzw[:, 1], want[1]) and np.array_equal(xyzw[:, 2], want[2])
