"""GPU: the reference's second tutorial, cudaParticlesPimpleFoam/TJunction (248 000 hex cells of 1 mm, records 63 MB:
beyond L2), driven the way its solver drives the fragments: seed from the dictionary's box, then per Eulerian step a new
U and ``#include "advect.H"`` (ceil(deltaT / dt) Lagrangian cycles).  pimpleFoam's field is replaced by a closed-form
pulsating split flow (cases/tjunction.py)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tjunction_pimple_loop_bit_exact(oracle_libs):
    """D = 0: every particle equals the CPU statement bit for bit after every Eulerian step, through the host mirror of
    the fragments (seeding order of cuda/particles.cu:78-97, initial locate, sort, fused cycles between frames)."""
    from cudaparticlesfoam_amd.api import CudaParticles
    from cudaparticlesfoam_amd.cases import tjunction as tj
    mesh = tj.tjunction_mesh()
    centres, _ = mesh.cell_centres_volumes()
    cw = oracle_libs.CellWalk(); t = cw.build(mesh)
    d = dict(tj.PARTICLE_DICT, numParticles=200_000, diffusionCoeff=0.0, endTime=10.0, saveInterval=10 ** 9)
    time0 = d["startTime"]
    p = CudaParticles(mesh, tj.split_flow_u(mesh, centres, time0, u0=5.0), d)
    assert p.outOfDomain == 0
    xyzw, cell = p.particles()
    x, y, z, c = xyzw[:, 0].copy(), xyzw[:, 1].copy(), xyzw[:, 2].copy(), cell.copy()
    ref0 = cw.locate_initial(x.copy(), y.copy(), z.copy(), t, nthreads=cw.max_threads)
    assert np.array_equal(ref0, cell)
    for e in range(6):
        now = time0 + e * tj.EULERIAN_DT
        U = tj.split_flow_u(mesh, centres, now, u0=5.0)
        n_cycles = max(int(math.ceil(tj.EULERIAN_DT / d["dt"])), 1)
        assert p.advect(now, tj.EULERIAN_DT, U=U) == n_cycles
        cw.step(x, y, z, c, tj.EULERIAN_DT / n_cycles, n_cycles, t, U, nthreads=cw.max_threads)
        xyzw, cell = p.particles()
        assert np.array_equal(cell, c), "cells differ after Eulerian step %d" % e
        assert np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) and np.array_equal(xyzw[:, 2], z)
    assert (c != ref0).mean() > 0.5              # most particles have left their first cell
    p.close()


def test_tjunction_tutorial_size_with_diffusion():
    """The dictionary as shipped: 4e6 particles, D = 1.5e-5, a frame every second cycle.  Nobody is lost (every boundary
    face reflects, like in the reference, which hands the walk no patch types), everybody stays inside the T, the frame
    cadence is the fragment's, and the cloud has moved down the duct by about u * t."""
    from cudaparticlesfoam_amd.api import CudaParticles
    from cudaparticlesfoam_amd.cases import tjunction as tj
    mesh = tj.tjunction_mesh()
    centres, _ = mesh.cell_centres_volumes()
    d = dict(tj.PARTICLE_DICT, endTime=10.0)
    frames = []
    p = CudaParticles(mesh, tj.split_flow_u(mesh, centres, d["startTime"]), d,
                      writer=lambda frame, xyzw, vel, cell: frames.append((frame, float(xyzw[:, 0].mean()), int((cell >= 0).sum()))))
    p.ctx.set_option("stats", 1)
    assert p.numParticles == 4_000_000 and p.outOfDomain == 0
    x0 = frames[0][1]
    done = 0
    for e in range(2):
        now = d["startTime"] + e * tj.EULERIAN_DT
        done += p.advect(now, tj.EULERIAN_DT, U=tj.split_flow_u(mesh, centres, now))
    assert done == 20
    assert [f[0] for f in frames] == [0] + list(range(1, 21, 2))          # step % 2 == 0 -> frame step + 1
    assert all(f[2] == 4_000_000 for f in frames)
    xyzw, cell = p.particles()
    assert (cell >= 0).all()
    lo, hi = mesh.bounds()
    assert (xyzw[:, :3] >= lo - 1e-12).all() and (xyzw[:, :3] <= hi + 1e-12).all()
    inside_t = (xyzw[:, 0] <= 0.2 + 1e-12) & (np.abs(xyzw[:, 1]) <= 0.01 + 1e-12) | (xyzw[:, 0] >= 0.2 - 1e-12)
    assert inside_t.all()
    cnt = p.ctx.counters()
    assert cnt["particle_steps"] == 4_000_000 * 20 and cnt["lost"] == 0 and cnt["reflections"] > 0
    # bulk speed 3 m/s x 1.5 (profile) x ~0.44 (mean of the two parabolas) ~ 2 m/s for 2 ms
    assert 1e-3 < float(xyzw[:, 0].mean()) - x0 < 8e-3
    p.close()
