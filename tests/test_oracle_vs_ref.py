"""CPU: oracle/tetwalk.c against the reference's own functions compiled for CPU (oracle/_ref).
Skipped where oracle/_ref could not be built (no /root/reference and no prebuilt .so)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def libs(oracle_libs):
    if not oracle_libs.have_ref():
        pytest.skip("oracle/_ref not available")
    return oracle_libs.RefLib(), oracle_libs.TetWalk()


def test_box_mesh_tables_and_cycles_bitwise(libs):
    ref, tw = libs
    pos, tets = ref.box_mesh(6, 5, 4)
    a, b = ref.face_table(pos, tets), tw.face_table(pos, tets)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    rng = np.random.default_rng(2)
    tv = rng.normal(size=(tets.shape[0], 3))
    m = ref.tables(pos, tets, tv)
    n = 3000
    P0 = np.zeros((n, 4)); P0[:, :3] = rng.uniform([0, 0, 0], [6, 5, 4], size=(n, 3)); P0[:, 3] = 1
    ids0 = np.zeros(n, np.int32)
    for _ in range(12):                                   # walk from tet 0 to the containing tet (<=50 hops each)
        ref.bary_query(P0, ids0, m)
    ids1 = np.zeros(n, np.int32)
    for _ in range(12):
        tw.bary_query(P0, ids1, m)
    assert np.array_equal(ids0, ids1) and (ids0 >= 0).all()
    outs = []
    for lib in (ref, tw):
        P, ids, v, d = P0.copy(), ids0.copy(), np.zeros((n, 4)), np.zeros((n, 4))
        lib.cycles(P, ids, v, d, 0.25, 60, m, nthreads=2)
        outs.append((P, ids, v, d))
    for u, w in zip(*outs):
        assert np.array_equal(u, w)
    assert (outs[0][0][:, 3] == 1).all()


def test_init_particles_bitwise(libs):
    ref, tw = libs
    a = ref.init_particles(777, [-0.02, 0.025, 1e-4], [0.0, 0.0, -1e-4])
    b = tw.init_particles(777, [-0.02, 0.025, 1e-4], [0.0, 0.0, -1e-4], order=1)
    assert np.array_equal(a, b)


def test_pitzdaily_cycles_bitwise(libs, pitz):
    from oracle.tetmesh import poly_to_tets
    ref, tw = libs
    mesh, U = pitz["mesh"], pitz["U_analytic"]
    pos, tets, tcell, tu = poly_to_tets(mesh, pitz["centres"], U)
    assert tets.shape == (146700, 4)                      # 12 tets per hex (src/initCuda.H:64)
    a, b = ref.face_table(pos, tets), tw.face_table(pos, tets)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    assert a[0].shape[0] == 318410                        # tri facets of the tet mesh
    m = ref.tables(pos, tets, tu)
    pz = pitz["pz"]
    n = 1500
    xyz = pz.uniform_points(31, n, *pz.INLET_BOX)
    P0 = np.zeros((n, 4)); P0[:, :3] = xyz; P0[:, 3] = 1
    # start guess: any tet, then the reference's own fix-up walk (several rounds of <=50 hops)
    outs = []
    for lib in (ref, tw):
        ids = np.full(n, 12 * 100, np.int32)
        for _ in range(40):
            lib.bary_query(P0, ids, m)
        P, v, d = P0.copy(), np.zeros((n, 4)), np.zeros((n, 4))
        lib.cycles(P, ids, v, d, 1e-4, 400, m, nthreads=4)
        outs.append((P, ids, v))
    for u, w in zip(*outs):
        assert np.array_equal(u, w)
