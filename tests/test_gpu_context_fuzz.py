"""GPU: randomised call sequences on ONE context's own cloud (cpf_step with plain / fused / velocity-storing launches, the
automatic and explicit re-sorts -- which carry ids and stored velocities along --, new fields, sort options) against the CPU
checker: positions, cells and the last frame's velocities in particle-id order, bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FUSE, STORE_VEL = 4, 2


@pytest.mark.parametrize("seed", range(int(os.environ.get("CPF_FUZZ_CTX", "8"))))       # (a longer campaign: CPF_FUZZ_CTX=400)
def test_random_call_sequences_on_the_context_cloud(seed, gpu_ctx_factory, oracle_libs):
    from cudaparticlesfoam_amd.cases import box_mesh
    rng = np.random.default_rng(9000 + seed)
    mesh = box_mesh(*[int(v) for v in rng.integers(3, 9, size=3)])
    hi = np.asarray(mesh.points).max(axis=0)
    cw = oracle_libs.CellWalk(); t = cw.build(mesh)
    n = int(rng.integers(2000, 60000))
    xyz = rng.uniform([-0.4, 0, 0], hi, size=(n, 3))                       # x < 0: outside, frozen from the first step on
    U = rng.normal(size=(mesh.n_cells, 3)) * 0.5 + np.array([0.8, 0, 0])
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(U)
    ctx.set_particles(xyz); n_out = ctx.locate_initial()
    x, y, z = (xyz[:, k].copy() for k in range(3))
    c = cw.locate_initial(x, y, z, t, nthreads=cw.max_threads)
    assert n_out == int((c < 0).sum())
    vel = np.zeros((n, 3))
    have_frame = False
    for _ in range(int(rng.integers(6, 16))):
        op = int(rng.integers(0, 10))
        if op <= 5:
            k = int(rng.integers(1, 9)); dt = float(rng.choice([0.05, 0.2, 0.45]))
            flags = [0, FUSE, STORE_VEL, 0, FUSE, STORE_VEL | 0][op]
            ctx.step(dt, 0.0, k, flags)
            if flags & STORE_VEL:
                if k > 1:
                    cw.step(x, y, z, c, dt, k - 1, t, U, nthreads=cw.max_threads)
                vel[:] = 0.0                                               # (a particle that is not stepped has no velocity in a frame)
                live = c >= 0
                v = np.zeros((n, 3))
                cw.step(x, y, z, c, dt, 1, t, U, vel_out=v, nthreads=cw.max_threads)
                vel[live] = v[live]
                have_frame = True
            else:
                cw.step(x, y, z, c, dt, k, t, U, nthreads=cw.max_threads)
        elif op == 6:
            ctx.set_option("sort_interval", int(rng.choice([0, 1, 3, 7, 50])))
        elif op == 7:
            ctx.sort_by_cell()
        elif op == 8:
            U = rng.normal(size=(mesh.n_cells, 3)) * 0.5 + np.array([0.8, 0, 0])
            ctx.set_velocity(U)
        else:
            ctx.set_option("sort_method", int(rng.choice([0, 2])))
        if rng.integers(0, 3) == 0:
            xyzw, cell = ctx.get_particles()
            assert np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) and np.array_equal(xyzw[:, 2], z)
            assert np.array_equal(np.where(cell < 0, -1, cell), np.where(c < 0, -1, c))
    xyzw, cell, gv = ctx.get_particles(want_vel=True)
    assert np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) and np.array_equal(xyzw[:, 2], z)
    assert np.array_equal(np.where(cell < 0, -1, cell), np.where(c < 0, -1, c))
    assert np.array_equal(xyzw[:, 3], np.where(cell == -2, 0.0, 1.0))
    if have_frame:
        bad = np.flatnonzero((gv[:, :3] != vel).any(axis=1))
        assert bad.size == 0, (seed, bad.size, bad[:5], gv[bad[:3]], vel[bad[:3]], cell[bad[:5]])
