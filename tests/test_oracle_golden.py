"""CPU: the oracle (plain-C restatement of the reference algorithm) against the committed golden
vectors, which were produced by the reference's own functions (tests/golden/make_golden.py)."""
import hashlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")


def _digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _case_inputs(name, pitz):
    from cudaparticlesfoam_amd.cases import box_mesh
    g = np.load(os.path.join(G, name + ".npz"))
    if name.startswith("pitz"):
        mesh, centres = pitz["mesh"], pitz["centres"]
        U = pitz["U_uniform"] if name == "pitz_uniform" else pitz["U_analytic"]
    else:
        mesh = box_mesh(10, 9, 8)
        centres, _ = mesh.cell_centres_volumes()
        U = g["U"]
    assert str(g["inputs_sha256"]) == _digest(mesh.points, mesh.face_verts, mesh.owner, mesh.neighbour, U), \
        "synthetic inputs drifted from the ones the goldens were generated with"
    return g, mesh, centres, U


@pytest.mark.parametrize("name", ["pitz_uniform", "pitz_analytic", "box_random"])
def test_tetwalk_reproduces_reference_goldens_bitwise(name, pitz, oracle_libs):
    from oracle.tetmesh import poly_to_tets
    g, mesh, centres, U = _case_inputs(name, pitz)
    tw = oracle_libs.TetWalk()
    pos, tets, tcell, tu = poly_to_tets(mesh, centres, U)
    m = tw.tables(pos, tets, tu)
    n = g["xyz0"].shape[0]
    P = np.zeros((n, 4)); P[:, :3] = g["xyz0"]; P[:, 3] = 1
    ids = g["tet0"].copy(); vels = np.zeros((n, 4)); disps = np.zeros((n, 4))
    done = 0
    for k in g["checkpoints"]:
        tw.cycles(P, ids, vels, disps, float(g["dt"]), int(k) - done, m, nthreads=tw.max_threads)
        done = int(k)
        assert np.array_equal(P, g["P_%d" % k]) and np.array_equal(ids, g["tet_%d" % k])
        assert np.array_equal(vels, g["vel_%d" % k])


@pytest.mark.parametrize("name", ["pitz_uniform", "pitz_analytic", "box_random"])
def test_cellwalk_matches_reference_goldens(name, pitz, oracle_libs):
    """The polyhedral-cell formulation (what the HIP kernels implement) vs the reference's tet walk."""
    g, mesh, centres, U = _case_inputs(name, pitz)
    cw = oracle_libs.CellWalk()
    t = cw.build(mesh)
    x, y, z = (g["xyz0"][:, k].copy() for k in range(3))
    c = (g["tet0"] // 12).astype(np.int32)
    lo, hi = mesh.bounds(); L = float(np.linalg.norm(hi - lo))
    done = 0
    for k in g["checkpoints"]:
        cw.step(x, y, z, c, float(g["dt"]), int(k) - done, t, U, nthreads=cw.max_threads)
        done = int(k)
        P, tet = g["P_%d" % k], g["tet_%d" % k]
        rel = np.sqrt((x - P[:, 0]) ** 2 + (y - P[:, 1]) ** 2 + (z - P[:, 2]) ** 2) / L
        assert rel.max() <= 1e-5, "k=%d max rel %.3e" % (k, rel.max())
        assert np.array_equal(tet // 12, c)


def test_stage_by_stage_goldens(oracle_libs):
    """Each of the reference's four wrappers separately (cudaAdvect, convexTetQuery, convexWallReflect,
    cudaMoveParticles), including the -(startTet+1) wall encoding between locate and reflect."""
    from cudaparticlesfoam_amd.cases import box_mesh
    from oracle.tetmesh import poly_to_tets
    g = np.load(os.path.join(G, "stages_box.npz"))
    tw = oracle_libs.TetWalk()
    mesh = box_mesh(10, 9, 8)
    pos, tets, tcell, tu = poly_to_tets(mesh, None, g["U"])
    m = tw.tables(pos, tets, tu)
    n = g["xyz0"].shape[0]
    P = np.zeros((n, 4)); P[:, :3] = g["xyz0"]; P[:, 3] = 1
    ids = g["tet0"].copy(); vels = np.zeros((n, 4)); disps = np.zeros((n, 4))
    tw.advect(P, ids, vels, disps, float(g["dt"]), m)
    assert np.array_equal(P, g["adv_P"]) and np.array_equal(vels, g["adv_vel"]) and np.array_equal(disps, g["adv_disp"])
    tw.locate(P, ids, disps, m)
    assert np.array_equal(ids, g["loc_tet"]) and (ids < 0).sum() > 10
    wall = ids < 0
    assert np.array_equal(-ids[wall] - 1, g["tet0"][wall])        # 1-based negated START tet (ConvexQuery.cu:212)
    tw.reflect(P, ids, disps, vels, m)
    assert np.array_equal(P, g["ref_P"]) and np.array_equal(ids, g["ref_tet"])
    assert np.array_equal(disps, g["ref_disp"]) and np.array_equal(vels, g["ref_vel"])
    tw.move(P, disps)
    assert np.array_equal(P, g["mov_P"]) and np.array_equal(disps, g["mov_disp"])


def test_face_table_golden(oracle_libs):
    g = np.load(os.path.join(G, "face_table_box.npz"))
    f, tf, fi = oracle_libs.TetWalk().face_table(g["positions"], g["tets"])
    assert np.array_equal(f, g["facets"]) and np.array_equal(tf, g["tetfacets"]) and np.array_equal(fi, g["faceinfo"])
    # every facet has a front or a back tet; boundary sides are negative and numbered 1..nBoundary
    bd = np.sort(-fi[fi < 0])
    assert np.array_equal(bd, np.arange(1, bd.size + 1))


def test_seeding_golden(oracle_libs):
    g = np.load(os.path.join(G, "init_particles.npz"))
    tw = oracle_libs.TetWalk()
    P = tw.init_particles(g["P"].shape[0], g["lower"], g["upper"], order=1)
    assert np.array_equal(P, g["P"])
    # the tutorial's seeding box has inverted y/z bounds (cudaParticlesDict:25): positions still land inside it
    lo = np.minimum(g["lower"], g["upper"]); hi = np.maximum(g["lower"], g["upper"])
    assert (P[:, :3] >= lo - 1e-15).all() and (P[:, :3] <= hi + 1e-15).all() and (P[:, 3] == 1).all()


def test_vertex_velocity_golden(oracle_libs):
    """"VertexVelocity" mode (particleAdvectKernel, cuda/particles.cu:244-313): the plain-C restatement reproduces the
    reference's cycles bit for bit; the product's formulation on CELL ids (find the tet of the cell that holds P, then
    weigh like the reference) gives the same velocities -- identical bits wherever it picks the reference's tet."""
    from cudaparticlesfoam_amd.cases import box_mesh
    from oracle.tetmesh import poly_to_tets
    g = np.load(os.path.join(G, "vertex_box.npz"))
    mesh = box_mesh(10, 9, 8)
    assert str(g["inputs_sha256"]) == _digest(mesh.points, mesh.face_verts, mesh.owner, mesh.neighbour, g["vertex_U"])
    tw, cw = oracle_libs.TetWalk(), oracle_libs.CellWalk()
    pos, tets, tcell, _ = poly_to_tets(mesh, None, np.zeros((mesh.n_cells, 3)))
    m = tw.tables(pos, tets, np.zeros((tets.shape[0], 3)))
    n = g["xyz0"].shape[0]
    P = np.zeros((n, 4)); P[:, :3] = g["xyz0"]; P[:, 3] = 1
    ids = g["tet0"].copy(); vels = np.zeros((n, 4)); disps = np.zeros((n, 4))
    # the product's stage on cell ids, first cycle
    Pc = P.copy(); vc = np.zeros((n, 4)); dc = np.zeros((n, 4))
    cw.advect_vertex(Pc, (ids // 12).astype(np.int32), vc, dc, float(g["dt"]), tets, 12, pos, g["vertex_U"])
    same = (vc == g["adv_vel"]).all(1)
    assert same.mean() > 0.99 and np.abs(vc - g["adv_vel"]).max() < 1e-13 and np.abs(dc - g["adv_disp"]).max() < 1e-13
    done = 0
    for k in g["checkpoints"]:
        for c in range(int(k) - done):
            tw.advect_vertex(P, ids, vels, disps, float(g["dt"]), m, g["vertex_U"])
            if done == 0 and c == 0:
                assert np.array_equal(vels, g["adv_vel"]) and np.array_equal(disps, g["adv_disp"])
            tw.locate(P, ids, disps, m); tw.reflect(P, ids, disps, vels, m); tw.move(P, disps)
        done = int(k)
        assert np.array_equal(P, g["P_%d" % k]) and np.array_equal(ids, g["tet_%d" % k]) and np.array_equal(vels, g["vel_%d" % k])
