"""GPU: the cone locate of the "VertexVelocity" advect (option "vertex_fast", csrc/cpf_kernels.hip VertexField) gives the bits of
the evaluation of all tets -- the reference's rule as this library states it: the tet of the particle's cell whose smallest
barycentric weight is largest (cuda/particles.cu:244-313 on cell ids, include/cpf.h cpf_stage_advect_vertex).

The claim under test: where ONE tet holds the particle with every weight above the margin, no other tet of the cell can win;
where none does (on a tet's face, edge, the apex; outside the cell by a rounding or by a lot) the kernel evaluates all tets as
before.  So clouds are built to sit exactly on, and a few ulps / 1e-12 / 1e-9 / 1e-7 off, everything a fan of tets has:
the apex, the cell's corners, face centres, edge midpoints, the internal planes between neighbouring tets."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _meshes():
    from cudaparticlesfoam_amd.cases import box_mesh, pitzdaily as pz
    yield "box 10x9x8", box_mesh(10, 9, 8)
    yield "graded box", box_mesh(12, 7, 5, lower=(0.0, 0.0, 0.0), upper=(0.3, 0.05, 0.02), grading=(4.0, 0.3, 2.0))
    yield "pitzDaily", pz.pitzdaily_mesh()


def _adversarial_points(mesh, centres, pos, tets, rng, per_kind=4000):
    """(points, cells): points of chosen cells that lie on the structure of the cell's own tet fan, and a little off it."""
    nC = mesh.n_cells
    cells = rng.integers(0, nC, per_kind)
    t = tets.reshape(nC, -1, 4)
    k = rng.integers(0, t.shape[1], per_kind)
    tv = pos[t[cells, k]]                                      # (m, 4, 3): apex, B, C, D of one tet of the cell
    A, B, C, D = tv[:, 0], tv[:, 1], tv[:, 2], tv[:, 3]
    w = rng.dirichlet([1, 1, 1], per_kind)
    sets = [A,                                                 # the apex itself
            B, 0.5 * (B + C),                                  # a corner of the cell, an edge (or face-diagonal) midpoint
            (B + C + D) / 3.0,                                 # on the cell's boundary face
            A + 0.37 * (B - A),                                # on an edge all tets around it share
            A + w[:, :1] * 0.6 * (B - A) + w[:, 1:2] * 0.6 * (C - A),      # on the internal plane (A, B, C) between two tets
            0.25 * (A + B + C + D),                            # well inside
            A + 1e-14 * (B - A)]                               # a rounding away from the apex
    pts, cs = [], []
    for base in sets:
        for eps in (0.0, 1e-15, 1e-12, 1e-9, 1e-7, 1e-5):
            off = rng.normal(size=base.shape) * eps * np.abs(base).max()
            pts.append(base + off); cs.append(cells)
    return np.concatenate(pts), np.concatenate(cs).astype(np.int32)


@pytest.mark.parametrize("name,mesh", list(_meshes()), ids=[n for n, _ in _meshes()])
def test_cone_locate_equals_all_tets_bitwise(name, mesh, gpu_ctx_factory):
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.api import StagedCloud
    rng = np.random.default_rng(606)
    centres, _ = mesh.cell_centres_volumes()
    pos, tets = mesh.tet_decomposition(centres)
    vU = rng.normal(size=pos.shape)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(np.zeros((mesh.n_cells, 3)))
    ctx.set_tets(pos, tets, 12); ctx.set_vertex_velocity(vU)
    assert "(cone locate)" in ctx.step_kernel_name(0.0, L.STEP_VERTEX_VELOCITY)
    lo, hi = mesh.bounds()
    uni = rng.uniform(lo, hi, size=(200_000, 3))
    ctx.set_particles(uni); ctx.locate_initial()
    _, ucell = ctx.get_particles()
    adv, acell = _adversarial_points(mesh, centres, pos, tets, rng)
    P = np.concatenate([uni[ucell >= 0], adv]); cell = np.concatenate([ucell[ucell >= 0], acell]).astype(np.int32)
    n = P.shape[0]
    P4 = np.ones((n, 4)); P4[:, :3] = P
    out = {}
    sc = StagedCloud(ctx, n)
    try:
        for fast in (1, 0):
            ctx.set_option("vertex_fast", fast)
            assert ("(cone locate)" if fast else "(all tets)") in ctx.step_kernel_name(0.0, L.STEP_VERTEX_VELOCITY)
            sc.set(P4, cell)
            sc.cudaAdvect(0.01, "VertexVelocity")
            out[fast] = (sc.vels, sc.disps, sc.particles)
    finally:
        sc.close()
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a, b, equal_nan=True), name
    assert np.isfinite(out[1][0][:, :3]).all() and np.abs(out[1][0][:, :3]).max() > 0.1
    # ... and through the fused cycle: 12 cycles with reflection, both ways
    res = []
    for fast in (1, 0):
        ctx.set_option("vertex_fast", fast)
        ctx.set_particles(P, cell)
        ctx.step(0.004 * float((hi - lo).min()), 0.0, 12, L.STEP_VERTEX_VELOCITY)
        res.append(ctx.get_particles())
    assert np.array_equal(res[0][0], res[1][0], equal_nan=True) and np.array_equal(res[0][1], res[1][1])
    ctx.set_option("vertex_fast", 1)


def test_a_decomposition_that_is_not_a_fan_keeps_the_full_evaluation(gpu_ctx_factory):
    """cpf_set_tets admits the cone locate only where the tets of a cell cannot overlap: one apex, one orientation, solid angles
    adding up to 4 pi, base triangles closing up.  A decomposition with a turned tet, with a cell whose tets start at different vertices, or with a tet
    listed twice in place of another is stepped by the evaluation of all tets -- and says so."""
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.cases import box_mesh
    mesh = box_mesh(4, 3, 2)
    centres, _ = mesh.cell_centres_volumes()
    pos, tets = mesh.tet_decomposition(centres)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(np.zeros((mesh.n_cells, 3)))
    for what, edit in (("orientations", lambda t: t.__setitem__((5 * 12 + 3, slice(2, 4)), t[5 * 12 + 3, [3, 2]])),
                       ("first vertex", lambda t: t.__setitem__((7 * 12 + 1, slice(0, 2)), t[7 * 12 + 1, [1, 0]])),
                       ("closed surface", lambda t: t.__setitem__(2 * 12 + 4, t[2 * 12 + 5]))):
        t = tets.copy(); edit(t)
        ctx.set_tets(pos, t, 12); ctx.set_vertex_velocity(np.ones(pos.shape))
        name = ctx.step_kernel_name(0.0, L.STEP_VERTEX_VELOCITY)
        assert "(all tets: cell" in name and what in name, name
    ctx.set_tets(pos, tets, 12); ctx.set_vertex_velocity(np.ones(pos.shape))
    assert "(cone locate)" in ctx.step_kernel_name(0.0, L.STEP_VERTEX_VELOCITY)


def test_a_fan_of_24_tets_per_cell_takes_the_cone_locate_on_the_generic_kernel(gpu_ctx_factory):
    """Any fan about one apex is admitted, not only the reference's 12 tets per hex: here every quad face is cut into four
    triangles about its centre (24 tets a cell).  The streaming kernel's staged locate is built for twelve, so this cycle runs on
    step_kernel_vertex -- with the cone locate, and with the same bits as the evaluation of all 24 tets."""
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.cases import box_mesh
    mesh = box_mesh(6, 5, 4, lower=(0.0, 0.0, 0.0), upper=(0.6, 0.4, 0.2), grading=(2.0, 1.0, 0.5))
    centres, _ = mesh.cell_centres_volumes()
    fc, _ = mesh.face_centres_areas()
    off, faces = mesh.cell_faces()
    nP, nC = mesh.n_points, mesh.n_cells
    pos = np.concatenate([mesh.points, centres, fc])
    tets = []
    for c in range(nC):
        for f in faces[off[c]:off[c + 1]]:
            loop = mesh.face_verts[mesh.face_offsets[f]:mesh.face_offsets[f + 1]]
            if mesh.owner[f] != c:
                loop = loop[::-1]                                  # outward for this cell
            for k in range(len(loop)):
                tets.append((nP + c, nP + nC + f, loop[k], loop[(k + 1) % len(loop)]))
    tets = np.asarray(tets, np.int32)
    assert tets.shape[0] == 24 * nC
    rng = np.random.default_rng(24)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(np.zeros((nC, 3)))
    ctx.set_tets(pos, tets, 24); ctx.set_vertex_velocity(rng.normal(size=pos.shape))
    assert ctx.step_kernel_name(0.0, L.STEP_VERTEX_VELOCITY) == "cpf::step_kernel_vertex<false, true, false> (cone locate)"
    lo, hi = mesh.bounds()
    P = rng.uniform(lo, hi, size=(100_000, 3))
    out = []
    for fast in (1, 0):
        ctx.set_option("vertex_fast", fast)
        ctx.set_particles(P); ctx.locate_initial()
        ctx.step(0.002, 0.0, 10, L.STEP_VERTEX_VELOCITY)
        out.append(ctx.get_particles())
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert np.abs(out[0][0][:, :3] - P).max() > 1e-3
    ctx.set_option("vertex_fast", 1)
