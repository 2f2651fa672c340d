"""CPU: rank-direct ingest.  A mesh is cut into the pieces `decomposePar` would hand the ranks
(processor patches = boundary faces of the pieces) and stitched again by cpf_merge_mesh_parts; the result must be
the original mesh -- same cells, same faces in the same order with the same orientation and starting vertex (so
the walk's planes come out bit-identical), points merged exactly -- whatever the number of pieces."""
import numpy as np
import pytest

from cudaparticlesfoam_amd import _lib as L
from cudaparticlesfoam_amd.api import merge_mesh_parts
from cudaparticlesfoam_amd.cases import box_mesh, split_into_parts


def _faces_as_coordinates(mesh, which):
    fo, fv, pts = np.asarray(mesh.face_offsets), np.asarray(mesh.face_verts), np.asarray(mesh.points)
    return [tuple(map(tuple, pts[fv[fo[f]:fo[f + 1]]])) for f in which]


@pytest.mark.parametrize("n_parts", [1, 2, 3, 7])
def test_split_then_merge_gives_the_mesh_back(pitz, n_parts):
    mesh = pitz["mesh"]
    parts = split_into_parts(mesh, n_parts)
    assert sum(p.n_cells for p in parts) == mesh.n_cells
    if n_parts > 1:
        assert sum(p.n_points for p in parts) > mesh.n_points           # points on the cuts are duplicated ...
        assert sum(p.n_internal for p in parts) < mesh.n_internal       # ... and cut faces are boundary faces there
    merged = merge_mesh_parts(parts)
    assert merged.n_cells == mesh.n_cells and merged.n_points == mesh.n_points
    assert merged.n_internal == mesh.n_internal and merged.n_faces == mesh.n_faces
    # interior faces: same order (upper-triangular), same owner/neighbour, same vertex loop incl. its start
    ni = mesh.n_internal
    assert np.array_equal(merged.owner[:ni], mesh.owner[:ni]) and np.array_equal(merged.neighbour, mesh.neighbour)
    assert _faces_as_coordinates(merged, range(ni)) == _faces_as_coordinates(mesh, range(ni))
    # boundary faces: the same set (order is piece order), each with its owner and vertex loop
    want = sorted(zip(_faces_as_coordinates(mesh, range(ni, mesh.n_faces)), np.asarray(mesh.owner[ni:]).tolist()))
    got = sorted(zip(_faces_as_coordinates(merged, range(ni, merged.n_faces)), np.asarray(merged.owner[ni:]).tolist()))
    assert got == want


def test_merge_handles_64bit_labels_negative_zero_and_bad_input():
    mesh = box_mesh(3, 2, 2, lower=(-1.0, 0.0, 0.0), upper=(2.0, 1.0, 1.0))
    parts = split_into_parts(mesh, 2)
    for p in parts:
        p.owner = p.owner.astype(np.int64); p.neighbour = p.neighbour.astype(np.int64)
        p.face_offsets = p.face_offsets.astype(np.int64); p.face_verts = p.face_verts.astype(np.int64)
    parts[1].points = np.where(parts[1].points == 0.0, -0.0, parts[1].points)      # -0.0 == 0.0: still one point
    merged = merge_mesh_parts(parts)
    assert merged.n_points == mesh.n_points and merged.n_internal == mesh.n_internal
    parts[0].owner = parts[0].owner.copy(); parts[0].owner[0] = 99                 # out of range
    with pytest.raises(L.CpfError) as e:
        merge_mesh_parts(parts)
    assert e.value.status == L.CPF_ERR_MESH and "piece 0" in str(e.value) and "owner" in str(e.value)
    with pytest.raises(L.CpfError):
        merge_mesh_parts([])


def test_a_baffle_inside_a_piece_stays_a_wall():
    """Two coincident boundary faces owned by two different cells of ONE piece are a baffle (a zero-thickness wall),
    not a processor patch: the stitcher must keep both as boundary faces.  Faces that coincide across DIFFERENT
    pieces are the cuts of the decomposition and are joined."""
    from cudaparticlesfoam_amd.cases.polymesh import PolyMesh
    mesh = box_mesh(4, 1, 1)
    ni = mesh.n_internal
    # turn the interior face between cells 0 and 1 into a baffle: the same vertex loop twice, once per side
    fo, fv = np.asarray(mesh.face_offsets), np.asarray(mesh.face_verts)
    faces = [fv[fo[f]:fo[f + 1]] for f in range(mesh.n_faces)]
    k = next(f for f in range(ni) if {int(mesh.owner[f]), int(mesh.neighbour[f])} == {0, 1})
    keep = [f for f in range(ni) if f != k]
    new_faces = [faces[f] for f in keep] + [faces[f] for f in range(ni, mesh.n_faces)] + [faces[k], faces[k][::-1]]
    owner = [int(mesh.owner[f]) for f in keep] + [int(mesh.owner[f]) for f in range(ni, mesh.n_faces)] + [0, 1]
    neigh = [int(mesh.neighbour[f]) for f in keep]
    offs = np.concatenate([[0], np.cumsum([len(f) for f in new_faces])]).astype(np.int32)
    baffled = PolyMesh(points=mesh.points.copy(), face_offsets=offs, face_verts=np.concatenate(new_faces).astype(np.int32),
                       owner=np.asarray(owner, np.int32), neighbour=np.asarray(neigh, np.int32), n_cells=mesh.n_cells)
    merged = merge_mesh_parts([baffled])                       # one piece: nothing may be joined
    assert merged.n_internal == ni - 1 and merged.n_faces == mesh.n_faces + 1
    parts = split_into_parts(baffled, 2)                       # cells {0,1} | {2,3}: the baffle lies inside piece 0
    merged2 = merge_mesh_parts(parts)
    assert merged2.n_internal == ni - 1 and merged2.n_faces == mesh.n_faces + 1
    pairs = set(zip(np.asarray(merged2.owner[:merged2.n_internal]).tolist(), np.asarray(merged2.neighbour).tolist()))
    assert (0, 1) not in pairs and (1, 2) in pairs             # the cut between the pieces WAS re-joined


def _cell_signature(mesh):
    """per cell: the sorted list of its faces as (frozenset of vertex coordinates, the cell across it or -1) -- what the walk
    sees of a cell, whatever the face order, orientation or starting vertex"""
    fo, fv, pts = np.asarray(mesh.face_offsets), np.asarray(mesh.face_verts), np.asarray(mesh.points)
    own, nei, ni = np.asarray(mesh.owner), np.asarray(mesh.neighbour), mesh.n_internal
    sig = [[] for _ in range(mesh.n_cells)]
    for f in range(mesh.n_faces):
        key = frozenset(map(tuple, pts[fv[fo[f]:fo[f + 1]]]))
        if f < ni:
            sig[own[f]].append((key, int(nei[f]))); sig[nei[f]].append((key, int(own[f])))
        else:
            sig[own[f]].append((key, -1))
    return sig


@pytest.mark.parametrize("seed", range(6))
def test_arbitrary_decompositions_stitch_to_the_renumbered_mesh(seed, oracle_libs):
    """What scotch hands the ranks, not `simple`: every cell goes to a random piece (pieces that are not connected, pieces that
    touch every other piece, cuts along every face direction).  The stitched mesh numbers its cells piece by piece (global id =
    cells of the lower pieces + local id): it must be the original mesh under exactly that renumbering -- every cell with the
    same faces (as point sets) and the same cell across each --, have its interior faces upper-triangular, and carry particles
    along the same paths (the CPU checker on both: same cells, positions to rounding -- a face the stitcher had to orient
    the other way round has its plane from the reversed vertex loop)."""
    rng = np.random.default_rng(seed)
    mesh = box_mesh(*[int(v) for v in rng.integers(2, 7, size=3)])
    n_parts = int(rng.integers(2, 7))
    if seed % 2:
        cell_part = rng.integers(0, n_parts, size=mesh.n_cells)                    # salt and pepper
    else:                                                                          # blobs: nearest of n_parts random centres
        centres, _ = mesh.cell_centres_volumes()
        seeds_ = centres[rng.choice(mesh.n_cells, size=n_parts, replace=False)]
        cell_part = np.argmin(((centres[:, None, :] - seeds_[None, :, :]) ** 2).sum(axis=2), axis=1)
    cell_part[: n_parts] = np.arange(n_parts)                                      # (no empty piece)
    parts = split_into_parts(mesh, n_parts, cell_part)
    merged = merge_mesh_parts(parts)
    assert merged.n_cells == mesh.n_cells and merged.n_points == mesh.n_points
    assert merged.n_internal == mesh.n_internal and merged.n_faces == mesh.n_faces
    assert (np.asarray(merged.owner[: merged.n_internal]) < np.asarray(merged.neighbour)).all()
    # new id of an original cell: pieces one after the other, inside a piece by original id
    order = np.lexsort((np.arange(mesh.n_cells), cell_part))                       # order[new] = old
    new_of_old = np.empty(mesh.n_cells, np.int64); new_of_old[order] = np.arange(mesh.n_cells)
    want, got = _cell_signature(mesh), _cell_signature(merged)
    for old in range(mesh.n_cells):
        w = sorted(((sorted(k), -1 if c < 0 else int(new_of_old[c])) for k, c in want[old]))
        g = sorted(((sorted(k), c) for k, c in got[new_of_old[old]]))
        assert w == g, (seed, old)
    # the same particles through both meshes
    cw = oracle_libs.CellWalk()
    t0, t1 = cw.build(mesh), cw.build(merged)
    U = rng.normal(size=(mesh.n_cells, 3)) * 0.7
    U1 = U[order]
    hi = np.asarray(mesh.points).max(axis=0)
    xyz = rng.uniform([0, 0, 0], hi, size=(3000, 3))
    a = [xyz[:, k].copy() for k in range(3)]; b = [xyz[:, k].copy() for k in range(3)]
    ca, cb = cw.locate_initial(*a, t0), cw.locate_initial(*b, t1)
    assert np.array_equal(new_of_old[ca], cb)
    cw.step(*a, ca, 0.3, 12, t0, U); cw.step(*b, cb, 0.3, 12, t1, U1)
    same_cell = new_of_old[ca] == cb
    assert same_cell.mean() > 0.995                     # (a particle within rounding of a face may end on its other side)
    for k in range(3):
        assert np.abs(a[k] - b[k])[same_cell].max() < 1e-11
