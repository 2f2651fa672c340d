"""CPU: the walk on meshes with many-faced cells (2:1 refinement) keeps every particle inside the cell it claims.

The reference cannot run such meshes (``src/initCuda.H:64``: ``tetsPerCell = 12``), so there is no golden vector to pin
them on; what the CPU statement must satisfy is the domain's own invariant.  A many-faced cell has COPLANAR faces (the
pieces of a split face), where the reference's plane-exit test is ambiguous twice over: the exit parameters of the pieces
tie, and a particle that came in through one piece sits on its siblings' plane a rounding error outside it, moving
inward.  ``oracle/cellwalk.c`` makes the coplanar faces of a cell ONE slot (cw_build: a face group) and settles the piece
at the exit point (trace_in_cell, resolve_group); with the reference's rule on the raw faces this test finds hundreds of
particles in the wrong cell after a single cycle (741 of 2e5 on the refined box).
"""
import numpy as np
import pytest


def worst_outside(t, xyz, cell):
    """max over each particle's claimed cell's faces of the signed plane distance (<= 0: inside)"""
    nf = np.diff(t.cell_off)
    pl = t.planes.reshape(-1, 4)
    worst = np.full(cell.shape[0], -np.inf)
    for k in range(int(nf.max())):
        has = nf[cell] > k
        p = pl[t.cell_off[cell[has]] + k]
        worst[has] = np.maximum(worst[has], p[:, 3] - (p[:, :3] * xyz[has]).sum(1))
    return worst


@pytest.mark.parametrize("D", [0.0, 0.4])
def test_refined_box_everybody_inside_the_claimed_cell(D, oracle_libs):
    from cudaparticlesfoam_amd.cases import refined_box
    mesh, _ = refined_box(8, 6, 5, (0, 0, 0), (8, 6, 5), ((2.0, 1.5, 1.0), (6.0, 4.5, 4.0)), grading=(2.0, 1.0, 0.5))
    cw = oracle_libs.CellWalk()
    t = cw.build(mesh)
    nf = np.diff(mesh.cell_faces()[0])                            # faces; the tables hold one slot per distinct plane
    assert nf.max() >= 18 and np.diff(t.cell_off).max() == 6 and t.n_groups == 86
    rng = np.random.default_rng(23)
    n = 100_000
    xyz = rng.uniform([0, 0, 0], [8, 6, 5], size=(n, 3))
    U = rng.normal(size=(mesh.n_cells, 3)) * 0.5
    x, y, z = (xyz[:, k].copy() for k in range(3))
    cell = cw.locate_initial(x, y, z, t, nthreads=cw.max_threads)
    assert (cell >= 0).all()
    gid = np.arange(n, dtype=np.int64)
    done = 0
    for cycles in (1, 5, 40):
        stats = cw.step(x, y, z, cell, 0.05, cycles - done, t, U, nthreads=cw.max_threads, D=D, gid=gid, step0=done, seed=5)
        done = cycles
        assert stats[2] == 0 and (cell >= 0).all()                # every boundary reflects: nobody is lost
        w = worst_outside(t, np.stack([x, y, z], 1), cell)
        assert w.max() <= 1e-9, (cycles, int((w > 1e-9).sum()), float(w.max()))
    assert (nf[cell] > 6).sum() > 1000 and stats[1] > 0


def test_refined_pitzdaily_everybody_inside_the_claimed_cell(oracle_libs, pitz):
    from cudaparticlesfoam_amd.cases import refined_pitzdaily
    pz = pitz["pz"]
    mesh, _ = refined_pitzdaily()
    cw = oracle_libs.CellWalk()
    t = cw.build(mesh)
    centres, _ = mesh.cell_centres_volumes()
    U = pz.analytic_step_u(mesh, centres)
    xyz = pz.uniform_points(99, 150_000, *pz.DOMAIN_BOX)
    x, y, z = (xyz[:, k].copy() for k in range(3))
    cell = cw.locate_initial(x, y, z, t, nthreads=cw.max_threads)
    keep = cell >= 0
    x, y, z, cell = x[keep].copy(), y[keep].copy(), z[keep].copy(), cell[keep].copy()
    cw.step(x, y, z, cell, 1e-4, 60, t, U, nthreads=cw.max_threads)
    alive = cell >= 0                                              # the outlet takes particles out: those are not "lost"
    w = worst_outside(t, np.stack([x, y, z], 1)[alive], cell[alive])
    assert alive.sum() > 100_000 and w.max() <= 1e-9, (int((w > 1e-9).sum()), float(w.max()))
