"""CPU: cells == the reference's own tets.  The cell-walk statement against the reference algorithm on the
reference's own test geometry, per-tet velocities, heavy wall reflection -- no decomposition in between."""
import numpy as np
import pytest

from tetcells import box_tets, tet_cell_polymesh


def test_box_tets_match_reference_generator(oracle_libs):
    if not oracle_libs.have_ref():
        pytest.skip("oracle/_ref not available")
    ref = oracle_libs.RefLib()
    pos, tets = box_tets(4, 3, 2)
    rp, rt = ref.box_mesh(4, 3, 2)
    assert np.array_equal(pos, rp) and np.array_equal(tets, rt)


def test_cellwalk_on_tet_cells_equals_reference_tet_walk(oracle_libs):
    tw, cw = oracle_libs.TetWalk(), oracle_libs.CellWalk()
    pos, tets = box_tets(6, 5, 4)
    mesh = tet_cell_polymesh(pos, tets)
    assert mesh.n_cells == tets.shape[0] and (np.diff(mesh.cell_faces()[0]) == 4).all()
    c, v = mesh.cell_centres_volumes()
    assert abs(v.sum() - 120.0) < 1e-9 and v.min() > 0
    rng = np.random.default_rng(4)
    U = rng.normal(size=(mesh.n_cells, 3))
    m = tw.tables(pos, tets, U)                          # per-tet velocity == per-cell velocity
    t = cw.build(mesh)
    n = 4000
    xyz = rng.uniform([0, 0, 0], [6, 5, 4], size=(n, 3))
    cell0 = cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t, nthreads=cw.max_threads)
    assert (cell0 >= 0).all()
    P = np.zeros((n, 4)); P[:, :3] = xyz; P[:, 3] = 1
    ids = cell0.copy()
    tw.bary_query(P, ids, m)                             # the reference's own fix-up agrees with the plane test
    assert np.array_equal(ids, cell0)
    x, y, z, cc = xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), cell0.copy()
    vels = np.zeros((n, 4)); disps = np.zeros((n, 4))
    L = float(np.linalg.norm([6, 5, 4]))
    refl = 0
    for k in (1, 9, 40):
        st = cw.step(x, y, z, cc, 0.2, k, t, U, nthreads=cw.max_threads)
        refl += int(st[1])
        tw.cycles(P, ids, vels, disps, 0.2, k, m, nthreads=tw.max_threads)
        rel = np.sqrt((x - P[:, 0]) ** 2 + (y - P[:, 1]) ** 2 + (z - P[:, 2]) ** 2) / L
        agree = (cc == ids) | ((cc < 0) & (ids < 0))
        assert rel.max() <= 1e-12 and agree.all(), (rel.max(), (~agree).sum())
    assert refl > 1000
