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
