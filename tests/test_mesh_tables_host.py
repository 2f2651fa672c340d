"""CPU: the product's mesh layer (csrc/cpf_mesh.cpp through the C-ABI's host-only entry cpf_build_mesh_tables_host) builds
the tables the CPU statement (oracle/cellwalk.c, cw_build) builds -- offsets, planes, neighbour codes and face groups, bit
for bit -- on hex meshes (the only ones the reference runs, src/initCuda.H:64: slots == faces there) and on meshes with
coplanar faces, prisms and true polyhedra.  The two are written independently; the GPU tests repeat the comparison on the
tables actually uploaded."""
import numpy as np
import pytest


def _same(mesh, oracle_libs):
    from cudaparticlesfoam_amd.api import build_mesh_tables_host
    t = oracle_libs.CellWalk().build(mesh)
    h = build_mesh_tables_host(mesh)
    assert np.array_equal(h["cell_off"], t.cell_off) and np.array_equal(h["nbr"], t.nbr)
    assert np.array_equal(h["planes"], t.planes.reshape(-1, 4))                 # bit-exact plane coefficients
    assert np.array_equal(h["group_off"], t.group_off) and np.array_equal(h["group_nbr"], t.group_nbr[:t.group_off[-1]])
    return t


def test_pitzdaily_and_tjunction_have_no_groups(oracle_libs, pitz):
    from cudaparticlesfoam_amd.cases import tjunction as tj
    t = _same(pitz["mesh"], oracle_libs)
    assert t.n_groups == 0 and t.cell_off[-1] == 49180 + 24170                   # slots == faces + internal faces
    t = _same(tj.tjunction_mesh(), oracle_libs)                                  # 248 000 cells, multi-block
    assert t.n_groups == 0 and t.cell_off[-1] == 6 * 248000


def test_refined_meshes_merge_coplanar_pieces(oracle_libs):
    from cudaparticlesfoam_amd.cases import refined_box, refined_pitzdaily
    mesh, _ = refined_box(8, 6, 5, (0, 0, 0), (8, 6, 5), ((2.0, 1.5, 1.0), (6.0, 4.5, 4.0)), grading=(2.0, 1.0, 0.5))
    t = _same(mesh, oracle_libs)
    assert np.diff(mesh.cell_faces()[0]).max() >= 18 and np.diff(t.cell_off).max() == 6 and t.n_groups == 86
    assert set(np.diff(t.group_off)) == {4}                                      # a face split 2 x 2: four pieces
    g = t.nbr[t.nbr < -(1 << 30)] - t.GROUP_BASE
    assert np.array_equal(np.sort(g), np.arange(t.n_groups))                     # every group is some slot's neighbour code, once
    mesh, _ = refined_pitzdaily()
    t = _same(mesh, oracle_libs)
    assert t.n_groups == 87 and set(np.diff(t.group_off)) == {2} and set(np.diff(t.cell_off)) == {6}


@pytest.mark.parametrize("every", [1, 4])
def test_cut_corner_grid_keeps_seven_slot_cells(every, oracle_libs):
    from cudaparticlesfoam_amd.cases.polygons import cut_corner_box
    mesh, kinds = cut_corner_box(9, 7, 2, every=every)
    t = _same(mesh, oracle_libs)
    slots = np.diff(t.cell_off)
    assert (slots == 7).sum() == 2 * kinds["pentagons"] and (slots == 5).sum() == 2 * kinds["triangles"]
    assert t.n_groups >= 2 * kinds["squares_with_hanging_node"]


def test_a_mesh_the_product_refuses_is_refused_here_too():
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.api import build_mesh_tables_host
    from cudaparticlesfoam_amd.cases import box_mesh
    m = box_mesh(2, 2, 2)
    m.owner = m.owner.copy(); m.owner[0] = 99                                     # out of range
    with pytest.raises(L.CpfError):
        build_mesh_tables_host(m)


def test_rounding_noise_in_face_normals_is_zero_and_pitzdaily_is_z_layered(pitz):
    """A component of a unit face normal of magnitude <= 1e-12 is stored as zero (csrc/cpf_mesh.cpp, restated in
    oracle/cellwalk.c): 35 % of pitzDaily's faces carry such noise out of the cross products, and with it in place neither
    the z-pair skip nor the zero-denominator skips of the walk can fire.  After it every cell of the mesh has its two
    faces with an exactly z-parallel unit normal in slots 4 and 5."""
    from cudaparticlesfoam_amd.api import build_mesh_tables_host
    h = build_mesh_tables_host(pitz["mesh"])
    pl = h["planes"].reshape(-1, 6, 4)
    n = pl[:, :, :3]
    assert not ((np.abs(n) <= 1e-12) & (n != 0)).any()
    assert np.allclose((n ** 2).sum(2), 1.0, rtol=0, atol=4e-16)
    isz = (n[:, :, 0] == 0) & (n[:, :, 1] == 0)
    assert isz[:, 4:].all() and not isz[:, :4].any() and (np.abs(n[:, 4:, 2]) == 1.0).all()
    assert (h["nbr"].reshape(-1, 6)[:, 4:] < 0).all()                            # front / back: boundary faces, one cell thick
    zw = pl[:, 4:, 3] * pl[:, 4:, 2]                                             # the planes' z positions agree to rounding
    assert np.ptp(zw.min(1)) < 1e-15 and np.ptp(zw.max(1)) < 1e-15


def test_one_cell_thick_mesh_with_slanted_side_faces_is_not_z_thin(pitz):
    """fold_z (csrc/cpf_walk.h) mirrors a kicked end point about the front / back plane BEFORE the walk and may then leave
    both z faces out of the rounds; its argument needs the four side faces of every cell to have nz == 0 exactly.  A block
    that is one cell thick in z but whose back plane is shifted in x (slanted side faces, planar z faces with normals exactly
    along z) must keep its z faces in the walk: z_layered stays, z_thin does not."""
    from cudaparticlesfoam_amd.api import mesh_flags_host
    from cudaparticlesfoam_amd.cases import block_mesh, box_mesh
    f = mesh_flags_host(pitz["mesh"])
    assert f == dict(all_hex=1, z_layered=1, z_thin=1, mixed=0)
    assert mesh_flags_host(box_mesh(6, 5, 1))["z_thin"] == 1                     # straight side faces
    assert mesh_flags_host(box_mesh(6, 5, 2))["z_thin"] == 0                     # two cells thick: z faces are not all walls
    v = np.array([[0, 0, 0], [6, 0, 0], [6, 5, 0], [0, 5, 0], [0.5, 0, 1], [6.5, 0, 1], [6.5, 5, 1], [0.5, 5, 1]], float)
    sheared = block_mesh(v, [dict(hex=range(8), n=(6, 5, 1), simple=(1, 1, 1))])
    f = mesh_flags_host(sheared)
    assert f["all_hex"] == 1 and f["z_layered"] == 1 and f["z_thin"] == 0


@pytest.mark.parametrize("cuts,want", [(1, 10), (2, 14)])
def test_chamfered_grid_has_ten_and_fourteen_slot_cells(cuts, want, oracle_libs):
    """Octagonal prisms: ten distinct planes (two cell records in the streaming kernel); dodecagonal prisms: fourteen (header
    record).  Their edge neighbours carry two hanging nodes on the shared edge: a face group of three coplanar pieces."""
    from cudaparticlesfoam_amd.api import mesh_flags_host
    from cudaparticlesfoam_amd.cases.polygons import chamfered_box
    mesh, kinds = chamfered_box(9, 6, 2, cuts)
    t = _same(mesh, oracle_libs)
    slots = np.diff(t.cell_off)
    assert set(slots) == {5, 6, want} and (slots == want).sum() == 2 * kinds["polygons"]
    assert set(np.diff(t.group_off)) == {3} and t.n_groups == 2 * 4 * kinds["polygons"]
    assert mesh_flags_host(mesh)["mixed"] == 2
    _, vol = mesh.cell_centres_volumes()
    assert abs(vol.sum() - 9 * 6 * 2) < 1e-9 and vol.min() > 0


@pytest.mark.parametrize("period,slot_set", [(6, {6, 7}), (1, {6, 7, 8, 10})])
def test_conformal_diamond_meshes_have_no_groups(period, slot_set, oracle_libs):
    """cases/polygons.py, diamond_box: true polyhedra without hanging nodes (no face groups): pentagonal prisms around a
    diamond (period 6), the truncated square tiling of octagonal prisms and diamonds (period 1)."""
    from cudaparticlesfoam_amd.cases.polygons import diamond_box
    mesh, kinds = diamond_box(12, 12, 2, period)
    t = _same(mesh, oracle_libs)
    assert t.n_groups == 0 and set(np.diff(t.cell_off)) == slot_set
    _, vol = mesh.cell_centres_volumes()
    assert abs(vol.sum() - 288.0) < 1e-9 and vol.min() > 0


def test_box_records_host():
    """Box records (csrc/cpf_walk.h "box records") against the full tables: every canonical slot k = 2 * axis + (n_a == -1) holds
    the offset, the neighbour and the place in the walk's own slot order of the plane with that normal; a mesh with one sloped
    face has none."""
    from cudaparticlesfoam_amd.api import build_mesh_tables_host, mesh_box_records_host
    from cudaparticlesfoam_amd.cases import box_mesh, pitzdaily
    mesh = box_mesh(5, 4, 3, lower=(-1.0, 0.5, 0.0), upper=(1.5, 2.0, 0.9), grading=(2.0, 1.0, 0.25))
    rec = mesh_box_records_host(mesh)
    t = build_mesh_tables_host(mesh)
    assert rec is not None and rec.shape == (mesh.n_cells, 16)
    nb = rec.view(np.int32).reshape(mesh.n_cells, 32)
    for c in range(mesh.n_cells):
        s0 = t["cell_off"][c]
        assert t["cell_off"][c + 1] - s0 == 6
        code = int(nb[c, 18]) & 0xFFFFFFFF
        for s in range(6):
            pl = t["planes"][s0 + s]
            axis = int(np.argmax(np.abs(pl[:3])))
            assert abs(pl[axis]) == 1.0 and np.count_nonzero(pl[:3]) == 1
            k = 2 * axis + (1 if pl[axis] < 0 else 0)
            assert rec[c, k] == pl[3] and nb[c, 12 + k] == t["nbr"][s0 + s]
            assert (code >> (3 * k)) & 7 == s
            others = [a for a in range(3) if a != axis]
            assert (code >> (18 + 2 * k)) & 1 == int(np.signbit(pl[others[0]]))
            assert (code >> (19 + 2 * k)) & 1 == int(np.signbit(pl[others[1]]))
        assert (rec[c, 10:13] == 0).all()
    assert mesh_box_records_host(pitzdaily.pitzdaily_mesh()) is None
    pts = np.array(mesh.points, dtype=np.float64)
    pts[np.argmax(pts.sum(1))] += np.array([0.0, 0.0, 0.01])          # one corner lifted: its three faces are no longer axis-aligned
    import dataclasses
    assert mesh_box_records_host(dataclasses.replace(mesh, points=pts)) is None
