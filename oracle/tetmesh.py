"""TEST INFRASTRUCTURE ONLY (oracle).  polyMesh -> tet mesh exactly as the reference's
init fragment builds it, so the SAME case can be fed to the reference's tet walk
(oracle/_ref, oracle/tetwalk.c) and to the product's polyhedral-cell walk.

Restates ``/root/reference/src/initCuda.H:76-124``:
  * tet k of cell c = (nPoints + c, tri[0], tri[1], tri[2])          (initCuda.H:99-105)
  * one velocity per tet = U[c]                                       (initCuda.H:106-108)
  * positions = mesh.points() ++ mesh.C()                             (initCuda.H:112-124)
and the OpenFOAM v2106 pieces it calls, which are NOT in the reference tree
(un-vendored dependency; semantics restated from OpenFOAM's published source,
"parity unpinned" at this boundary -- SURVEY.md 8c):
  * ``polyMeshTetDecomposition::cellTetIndices``: for each face of ``mesh.cells()[c]``,
    for tetPt = 1..nVerts-2 one tet;
  * ``tetIndices::faceTriIs``: base = f[tetBasePtIs[face]] (0 here), a = f[(tetPt+base) % n],
    b = f[next(a)], swapped when the cell is not the face owner, so (base,a,b) is
    outward-oriented for the cell.
For planar-faced convex cells the containing CELL (hence U, hence positions) does not
depend on which valid fan decomposition is used.
"""
from __future__ import annotations

import numpy as np


def poly_to_tets(mesh, cell_centres=None, cell_u=None):
    """Returns (positions (nP+nC,3) f64, tets (nT,4) i32, tet_cell (nT,) i32, tet_u (nT,3) f64|None)."""
    if cell_centres is None:
        cell_centres, _ = mesh.cell_centres_volumes()
    nP = mesh.n_points
    coff, cfaces = mesh.cell_faces()
    fo = mesh.face_offsets.astype(np.int64)
    nv = np.diff(fo)
    cell_of_slot = np.repeat(np.arange(mesh.n_cells), np.diff(coff))
    f = cfaces.astype(np.int64)
    ntri = nv[f] - 2                                   # tets contributed by each (cell, face) slot
    slot = np.repeat(np.arange(f.size), ntri)
    first = np.cumsum(ntri) - ntri
    tetpt = np.arange(slot.size) - np.repeat(first, ntri) + 1   # 1..n-2
    face = f[slot]; cell = cell_of_slot[slot]
    n = nv[face]
    base_i = np.zeros_like(face)                        # tetBasePtIs == 0
    a_i = (tetpt + base_i) % n
    b_i = (a_i + 1) % n
    not_owner = mesh.owner[face] != cell
    a_i, b_i = np.where(not_owner, b_i, a_i), np.where(not_owner, a_i, b_i)
    fv = mesh.face_verts
    tets = np.stack([nP + cell, fv[fo[face] + base_i], fv[fo[face] + a_i], fv[fo[face] + b_i]], 1).astype(np.int32)
    positions = np.ascontiguousarray(np.concatenate([mesh.points, cell_centres]), dtype=np.float64)
    tet_u = None if cell_u is None else np.ascontiguousarray(np.asarray(cell_u, dtype=np.float64)[cell])
    return positions, tets, cell.astype(np.int32), tet_u
