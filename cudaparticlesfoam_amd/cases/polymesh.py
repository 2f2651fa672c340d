"""OpenFOAM-style ``polyMesh`` container (points, face-vertex lists, owner, neighbour).

This is the data the reference's fragments read from ``mesh`` before they build
their tet mesh (``/root/reference/src/initCuda.H:76-124``: ``mesh.cells()``,
``mesh.points()``, ``mesh.C()``) and exactly what crosses this repo's C-ABI
(``include/cpf.h`` ``cpf_set_mesh``).  Conventions are OpenFOAM's [OpenFOAM, not
in the reference tree]:

* faces ``0..n_internal-1`` are internal, ordered by (owner, neighbour) with
  ``owner < neighbour`` ("upper-triangular order"); boundary faces follow,
  grouped by patch;
* a face's vertex loop is counter-clockwise seen from outside its owner cell,
  i.e. the face normal points owner -> neighbour (outward on the boundary);
* a cell's face list (``mesh.cells()[c]``) = the faces it owns in ascending
  face id, then the faces it is neighbour of in ascending face id
  (``primitiveMesh::calcCells``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np


@dataclass
class PolyMesh:
    points: np.ndarray          # (nPoints, 3) float64
    face_offsets: np.ndarray    # (nFaces+1,) int32  CSR offsets into face_verts
    face_verts: np.ndarray      # (sum nVerts,) int32
    owner: np.ndarray           # (nFaces,) int32
    neighbour: np.ndarray       # (nInternalFaces,) int32
    n_cells: int
    patches: List[Tuple[str, str, int, int]] = field(default_factory=list)  # (name, type, start, size)

    # ------------------------------------------------------------------ sizes
    @property
    def n_points(self) -> int:
        return int(self.points.shape[0])

    @property
    def n_faces(self) -> int:
        return int(self.owner.shape[0])

    @property
    def n_internal(self) -> int:
        return int(self.neighbour.shape[0])

    # ------------------------------------------------------- derived topology
    def cell_faces(self) -> Tuple[np.ndarray, np.ndarray]:
        """CSR (offsets, faces) of ``mesh.cells()`` in ``primitiveMesh::calcCells`` order."""
        nC = self.n_cells
        own = self.owner.astype(np.int64)
        nei = self.neighbour.astype(np.int64)
        counts = np.bincount(own, minlength=nC) + np.bincount(nei, minlength=nC)
        offsets = np.zeros(nC + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        faces = np.empty(int(offsets[-1]), dtype=np.int32)
        own_counts = np.bincount(own, minlength=nC)
        # owned faces, ascending face id (stable sort by owner keeps face order)
        order_o = np.argsort(own, kind="stable")
        start_o = offsets[:-1]
        rank_o = np.arange(own.size) - np.repeat(np.cumsum(own_counts) - own_counts, own_counts)
        faces[start_o[own[order_o]] + rank_o] = order_o.astype(np.int32)
        # neighbour faces after them
        nei_counts = np.bincount(nei, minlength=nC)
        order_n = np.argsort(nei, kind="stable")
        rank_n = np.arange(nei.size) - np.repeat(np.cumsum(nei_counts) - nei_counts, nei_counts)
        faces[start_o[nei[order_n]] + own_counts[nei[order_n]] + rank_n] = order_n.astype(np.int32)
        return offsets.astype(np.int32), faces

    # --------------------------------------------------------------- geometry
    def face_centres_areas(self) -> Tuple[np.ndarray, np.ndarray]:
        """``mesh.faceCentres()``/``faceAreas()`` (primitiveMeshFaceCentresAndAreas):
        triangles about the vertex average, area-weighted centre."""
        nF = self.n_faces
        fo = self.face_offsets.astype(np.int64)
        nv = np.diff(fo)
        fid = np.repeat(np.arange(nF), nv)
        P = self.points[self.face_verts]
        est = np.zeros((nF, 3))
        np.add.at(est, fid, P)
        est /= nv[:, None]
        # next vertex within each face loop
        idx = np.arange(self.face_verts.size)
        nxt = idx + 1
        last = fo[1:] - 1
        nxt[last] = fo[:-1]
        Pn = self.points[self.face_verts[nxt]]
        c = P + Pn + est[fid]
        n = np.cross(Pn - P, est[fid] - P)
        a = np.sqrt((n * n).sum(1))
        sumN = np.zeros((nF, 3)); np.add.at(sumN, fid, n)
        sumA = np.zeros(nF); np.add.at(sumA, fid, a)
        sumAc = np.zeros((nF, 3)); np.add.at(sumAc, fid, a[:, None] * c)
        tri = nv == 3
        ctr = np.where((sumA > 0)[:, None], (1.0 / 3.0) * sumAc / np.maximum(sumA, 1e-300)[:, None], est)
        area = 0.5 * sumN
        if tri.any():
            f = np.nonzero(tri)[0]
            p0 = self.points[self.face_verts[fo[f]]]
            p1 = self.points[self.face_verts[fo[f] + 1]]
            p2 = self.points[self.face_verts[fo[f] + 2]]
            ctr[f] = (1.0 / 3.0) * (p0 + p1 + p2)
            area[f] = 0.5 * np.cross(p1 - p0, p2 - p0)
        return ctr, area

    def cell_centres_volumes(self) -> Tuple[np.ndarray, np.ndarray]:
        """``mesh.C()``/``mesh.V()`` (primitiveMeshCellCentresAndVols): face-centre
        average as estimate, then pyramid-volume-weighted centroid."""
        fc, fa = self.face_centres_areas()
        nC = self.n_cells
        own = self.owner.astype(np.int64)
        nei = self.neighbour.astype(np.int64)
        nI = self.n_internal
        est = np.zeros((nC, 3)); cnt = np.zeros(nC)
        np.add.at(est, own, fc); np.add.at(cnt, own, 1)
        np.add.at(est, nei, fc[:nI]); np.add.at(cnt, nei, 1)
        est /= cnt[:, None]
        ctr = np.zeros((nC, 3)); vol = np.zeros(nC)
        # owner side
        pyr = (fa * (fc - est[own])).sum(1)
        pc = 0.75 * fc + 0.25 * est[own]
        np.add.at(ctr, own, pyr[:, None] * pc); np.add.at(vol, own, pyr)
        # neighbour side (area vector points away from owner => into neighbour)
        pyr = (fa[:nI] * (est[nei] - fc[:nI])).sum(1)
        pc = 0.75 * fc[:nI] + 0.25 * est[nei]
        np.add.at(ctr, nei, pyr[:, None] * pc); np.add.at(vol, nei, pyr)
        ctr = np.where((np.abs(vol) > 1e-300)[:, None], ctr / np.where(vol == 0, 1, vol)[:, None], est)
        return ctr, vol / 3.0

    def tet_decomposition(self, cell_centres: np.ndarray | None = None) -> Tuple[np.ndarray, np.ndarray]:
        """What the reference's init fragment hands its library for the "VertexVelocity" mode (``src/initCuda.H:86-124``):
        tet-mesh vertices = ``mesh.points()`` ++ ``mesh.C()`` and, per cell in ``mesh.cells()`` order, one tet per face
        triangle -- apex = the cell's centre vertex, base = the fan (f[0], f[k], f[k+1]), k = 1 .. n-2, of each face, turned
        so that it is seen counter-clockwise from outside the cell (``tetIndices::faceTriIs`` with base point 0).  A hex
        gives 12 tets.  Returns (positions [nPoints + nCells][3], tets [nTets][4] int32) for ``cpf_set_tets``; every cell
        must yield the same number of tets (the reference: hexes only)."""
        if cell_centres is None:
            cell_centres, _ = self.cell_centres_volumes()
        off, faces = self.cell_faces()
        fo = self.face_offsets.astype(np.int64)
        fv = self.face_verts.astype(np.int64)
        tets = []
        cells_of_slots = np.repeat(np.arange(self.n_cells, dtype=np.int64), np.diff(off))
        face_sizes = np.diff(fo)[faces]
        for nverts in np.unique(face_sizes):                      # faces of one size at a time: a dense (slots, n) vertex table
            sel = np.nonzero(face_sizes == nverts)[0]
            f = faces[sel].astype(np.int64)
            loops = fv[fo[f][:, None] + np.arange(nverts)[None, :]]
            inward = self.owner[f] != cells_of_slots[sel]           # the loop is outward for the owner: turn it for the neighbour
            for k in range(1, int(nverts) - 1):
                a, b = loops[:, k], loops[:, k + 1]
                a, b = np.where(inward, b, a), np.where(inward, a, b)
                tets.append(np.stack([sel, np.full(sel.size, k), self.n_points + cells_of_slots[sel], loops[:, 0], a, b], 1))
        t = np.concatenate(tets)
        t = t[np.lexsort((t[:, 1], t[:, 0]))]                      # cell order, then the cell's face order, then the fan
        return (np.ascontiguousarray(np.concatenate([self.points, cell_centres]), dtype=np.float64),
                np.ascontiguousarray(t[:, 2:], dtype=np.int32))

    def bounds(self) -> Tuple[np.ndarray, np.ndarray]:
        return self.points.min(0), self.points.max(0)

    # ------------------------------------------------------------- reordering
    def renumber_cells(self, new_of_old: np.ndarray) -> "PolyMesh":
        """Return the same mesh with cell ``c`` renamed ``new_of_old[c]`` and faces
        re-sorted/flipped to keep OpenFOAM's conventions (what ``renumberMesh``
        does).  Used to give ranks contiguous x-slabs of cells."""
        new_of_old = np.asarray(new_of_old, dtype=np.int64)
        nI = self.n_internal
        fo = self.face_offsets.astype(np.int64)
        own = new_of_old[self.owner.astype(np.int64)]
        nei = new_of_old[self.neighbour.astype(np.int64)]
        flip = np.zeros(self.n_faces, dtype=bool)
        flip[:nI] = own[:nI] > nei
        o2 = own.copy(); n2 = nei.copy()
        o2[:nI] = np.where(flip[:nI], nei, own[:nI]); n2 = np.where(flip[:nI], own[:nI], nei)
        order_int = np.lexsort((n2, o2[:nI]))
        pieces = [order_int]
        patches = []
        pos = nI
        for (name, typ, start, size) in self.patches:
            f = np.arange(start, start + size)
            f = f[np.argsort(o2[f], kind="stable")]
            pieces.append(f); patches.append((name, typ, pos, size)); pos += size
        if not self.patches:
            f = np.arange(nI, self.n_faces)
            pieces.append(f[np.argsort(o2[f], kind="stable")])
        order = np.concatenate(pieces)
        new_verts = []; new_off = [0]
        for f in order:
            v = self.face_verts[fo[f]:fo[f + 1]]
            if flip[f]:
                v = np.concatenate(([v[0]], v[:0:-1]))
            new_verts.append(v); new_off.append(new_off[-1] + v.size)
        return PolyMesh(self.points.copy(), np.asarray(new_off, dtype=np.int32),
                        np.concatenate(new_verts).astype(np.int32), o2[order].astype(np.int32),
                        n2[order_int].astype(np.int32), self.n_cells, patches)


def build_polymesh_from_cells(points: np.ndarray, cell_face_loops: List[List[Tuple[int, ...]]],
                              boundary_patch_of: Dict[Tuple[int, ...], int] | None = None,
                              patch_names: List[Tuple[str, str]] | None = None) -> PolyMesh:
    """Generic cell->faces assembler: ``cell_face_loops[c]`` lists each face of cell
    ``c`` as a vertex loop that is counter-clockwise seen from OUTSIDE the cell."""
    face_of_key: Dict[Tuple[int, ...], int] = {}
    loops: List[Tuple[int, ...]] = []
    own: List[int] = []
    nei: List[int] = []
    for c, fl in enumerate(cell_face_loops):
        for loop in fl:
            key = tuple(sorted(loop))
            f = face_of_key.get(key)
            if f is None:
                face_of_key[key] = len(loops)
                loops.append(tuple(loop)); own.append(c); nei.append(-1)
            else:
                nei[f] = c
    own_a = np.asarray(own); nei_a = np.asarray(nei)
    internal = np.nonzero(nei_a >= 0)[0]
    boundary = np.nonzero(nei_a < 0)[0]
    order_int = internal[np.lexsort((nei_a[internal], own_a[internal]))]
    patches = []
    if boundary_patch_of is not None and patch_names is not None:
        pid = np.asarray([boundary_patch_of[tuple(sorted(loops[f]))] for f in boundary])
        order_b = []
        pos = order_int.size
        for p, (name, typ) in enumerate(patch_names):
            fp = boundary[pid == p]
            fp = fp[np.argsort(own_a[fp], kind="stable")]
            order_b.append(fp); patches.append((name, typ, pos, int(fp.size))); pos += fp.size
        order_b = np.concatenate(order_b) if order_b else np.zeros(0, dtype=np.int64)
    else:
        order_b = boundary[np.argsort(own_a[boundary], kind="stable")]
    order = np.concatenate([order_int, order_b]).astype(np.int64)
    offs = [0]; verts: List[int] = []
    for f in order:
        verts.extend(loops[f]); offs.append(len(verts))
    return PolyMesh(np.ascontiguousarray(points, dtype=np.float64), np.asarray(offs, dtype=np.int32),
                    np.asarray(verts, dtype=np.int32), own_a[order].astype(np.int32),
                    nei_a[order_int].astype(np.int32), len(cell_face_loops), patches)


def split_into_parts(mesh: "PolyMesh", n_parts: int, cell_part=None) -> List["PolyMesh"]:
    """What ``decomposePar`` hands the ranks: piece p gets its cells with local cell ids (in the order of their global
    ids), its own point list, its interior faces, then its boundary faces -- the physical ones AND the faces cut by the
    decomposition (processor patches, oriented outward from the piece like any boundary face).  Input of the rank-direct
    ingest (cpf_merge_mesh_parts).  ``cell_part`` (optional, [n_cells] part ids): an arbitrary decomposition (what scotch
    gives); default: contiguous ranges ``[p*nC/n_parts, (p+1)*nC/n_parts)`` ("simple")."""
    nC = mesh.n_cells
    if cell_part is None:
        cuts = [(p * nC) // n_parts for p in range(n_parts + 1)]
        cell_part = np.zeros(nC, dtype=np.int64)
        for p in range(n_parts):
            cell_part[cuts[p]:cuts[p + 1]] = p
    cell_part = np.asarray(cell_part, dtype=np.int64)
    assert cell_part.shape == (nC,) and cell_part.min() >= 0 and cell_part.max() < n_parts
    local_cell = np.zeros(nC, dtype=np.int64)
    for p in range(n_parts):
        mine = np.nonzero(cell_part == p)[0]
        local_cell[mine] = np.arange(mine.size)
    own = np.asarray(mesh.owner, dtype=np.int64)
    nei = np.full(own.shape[0], -1, dtype=np.int64)
    nei[: mesh.n_internal] = np.asarray(mesh.neighbour, dtype=np.int64)
    own_part = cell_part[own]
    nei_part = np.where(nei >= 0, cell_part[np.maximum(nei, 0)], -1)
    fo = np.asarray(mesh.face_offsets, dtype=np.int64); fv = np.asarray(mesh.face_verts, dtype=np.int64)
    parts = []
    for p in range(n_parts):
        o_in = own_part == p
        n_in = nei_part == p
        interior = np.nonzero(o_in & n_in)[0]
        as_owner = np.nonzero(o_in & ~n_in)[0]                  # physical boundary or cut face, we are the owner
        as_neigh = np.nonzero(~o_in & n_in)[0]                  # cut face, we are the neighbour: flip it
        faces, owners, neighs = [], [], []
        for f in interior:
            faces.append(fv[fo[f]:fo[f + 1]]); owners.append(local_cell[own[f]]); neighs.append(local_cell[nei[f]])
        for f in as_owner:
            faces.append(fv[fo[f]:fo[f + 1]]); owners.append(local_cell[own[f]])
        for f in as_neigh:
            faces.append(fv[fo[f]:fo[f + 1]][::-1]); owners.append(local_cell[nei[f]])
        used = np.unique(np.concatenate(faces)) if faces else np.zeros(0, np.int64)
        local = np.full(mesh.n_points, -1, dtype=np.int64)
        local[used] = np.arange(used.size)
        offs = np.zeros(len(faces) + 1, dtype=np.int64)
        np.cumsum([len(f) for f in faces], out=offs[1:])
        verts = local[np.concatenate(faces)] if faces else np.zeros(0, np.int64)
        parts.append(PolyMesh(points=np.asarray(mesh.points)[used].copy(), face_offsets=offs.astype(np.int32),
                              face_verts=verts.astype(np.int32), owner=np.asarray(owners, dtype=np.int32),
                              neighbour=np.asarray(neighs, dtype=np.int32), n_cells=int((cell_part == p).sum())))
    return parts
