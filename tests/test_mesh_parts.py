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
