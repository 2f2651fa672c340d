"""A small multi-block structured hex mesher restating what ``blockMesh`` does for the
reference's tutorial cases (``.../pitzDaily/system/blockMeshDict:17-150`` and
``.../TJunction/system/blockMeshDict``): straight-edged hex blocks, per-edge
multi-section geometric grading, coincident block-interface points merged.

``blockMesh`` itself is OpenFOAM (not in the reference tree, not installed):
the grading rule is restated from its documentation and
point coordinates are NOT claimed to be bit-identical to OpenFOAM's -- only the
topology counts are (tests/test_cases.py).  Every consumer in this repo (oracle,
HIP kernels, CPU baseline) is fed the same arrays, so parity is unaffected.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .polymesh import PolyMesh

Grading = Sequence[Tuple[float, float, float]]  # (length fraction, cell fraction, expansion ratio) sections

# hexModel faces, vertex loops counter-clockwise seen from outside (x-,x+,y-,y+,z-,z+)
HEX_FACES = np.array([[0, 4, 7, 3], [1, 2, 6, 5], [0, 1, 5, 4], [3, 7, 6, 2], [0, 3, 2, 1], [4, 5, 6, 7]])
# the 12 block edges in blockMesh edgeGrading order: 0-3 along x, 4-7 along y, 8-11 along z,
# each given by the (other-two-coordinates) corner it sits at
_XEDGE_VW = [(0, 0), (1, 0), (1, 1), (0, 1)]   # edges 0..3 at (v,w)
_YEDGE_UW = [(0, 0), (1, 0), (1, 1), (0, 1)]   # edges 4..7 at (u,w)
_ZEDGE_UV = [(0, 0), (1, 0), (1, 1), (0, 1)]   # edges 8..11 at (u,v)


def as_grading(g) -> Grading:
    if isinstance(g, (int, float)):
        return [(1.0, 1.0, float(g))]
    return [tuple(map(float, s)) for s in g]


def line_divide(n: int, grading) -> np.ndarray:
    """Division points lambda_0=0..lambda_n=1 of one block edge (``lineDivide``)."""
    g = as_grading(grading)
    lf = np.array([s[0] for s in g]); cf = np.array([s[1] for s in g]); er = [s[2] for s in g]
    lf = lf / lf.sum(); cf = cf / cf.sum()
    divs = np.floor(cf * n + 0.5).astype(int)
    if divs.sum() != n:
        divs[int(np.argmax(cf))] += n - divs.sum()
    lam = np.zeros(n + 1)
    start = 1; frac0 = 0.0
    for s in range(len(g)):
        m = int(divs[s])
        if m > 0:
            i = np.arange(1, m + 1, dtype=np.float64)
            if er[s] == 1.0 or m == 1:
                lam[start:start + m] = frac0 + lf[s] * i / m
            else:
                r = er[s] ** (1.0 / (m - 1))
                lam[start:start + m] = frac0 + lf[s] * (1.0 - r ** i) / (1.0 - r ** m)
        frac0 += lf[s]; start += m
    lam[n] = 1.0
    return lam


def _block_points(corners: np.ndarray, n: Tuple[int, int, int], edge_grading: Sequence) -> np.ndarray:
    """Points of one block, shape (nz+1, ny+1, nx+1, 3).  Edge gradings are honoured
    exactly on the 12 edges; interior parameters solve the transfinite fixed point
    u = bilerp(x-edge params; v, w) etc., so a block face depends only on its own 4 edges."""
    nx, ny, nz = n
    s = [line_divide(nx, edge_grading[e]) for e in range(0, 4)]
    t = [line_divide(ny, edge_grading[e]) for e in range(4, 8)]
    r = [line_divide(nz, edge_grading[e]) for e in range(8, 12)]
    S = [a[None, None, :] for a in s]; T = [a[None, :, None] for a in t]; R = [a[:, None, None] for a in r]
    shape = (nz + 1, ny + 1, nx + 1)
    u = np.broadcast_to(sum(S) / 4.0, shape).copy()
    v = np.broadcast_to(sum(T) / 4.0, shape).copy()
    w = np.broadcast_to(sum(R) / 4.0, shape).copy()

    def bil(vals, a, b):  # vals at corners (0,0),(1,0),(1,1),(0,1)
        return (1 - a) * (1 - b) * vals[0] + a * (1 - b) * vals[1] + a * b * vals[2] + (1 - a) * b * vals[3]

    for _ in range(60):
        un = bil(S, v, w); un[..., 0] = 0.0; un[..., nx] = 1.0
        vn = bil(T, un, w); vn[:, 0, :] = 0.0; vn[:, ny, :] = 1.0
        wn = bil(R, un, vn); wn[0] = 0.0; wn[nz] = 1.0
        d = max(np.abs(un - u).max(), np.abs(vn - v).max(), np.abs(wn - w).max())
        u, v, w = un, vn, wn
        if d < 1e-16:
            break
    c = corners
    u_, v_, w_ = u[..., None], v[..., None], w[..., None]
    return ((1 - u_) * (1 - v_) * (1 - w_) * c[0] + u_ * (1 - v_) * (1 - w_) * c[1] + u_ * v_ * (1 - w_) * c[2]
            + (1 - u_) * v_ * (1 - w_) * c[3] + (1 - u_) * (1 - v_) * w_ * c[4] + u_ * (1 - v_) * w_ * c[5]
            + u_ * v_ * w_ * c[6] + (1 - u_) * v_ * w_ * c[7])


def hexes_to_polymesh(points: np.ndarray, hexes: np.ndarray,
                      classify: Optional[Callable[[np.ndarray, np.ndarray, np.ndarray], np.ndarray]] = None,
                      patch_names: Optional[List[Tuple[str, str]]] = None) -> PolyMesh:
    """Assemble OpenFOAM-ordered faces/owner/neighbour from hex cells (vectorised).
    ``classify(face_centres, owner_cells, local_face_slot) -> patch index`` for boundary faces
    (slots: 0 x-, 1 x+, 2 y-, 3 y+, 4 z-, 5 z+ in block-local axes)."""
    nC = hexes.shape[0]
    loops = hexes[:, HEX_FACES].reshape(nC * 6, 4)           # cell-major, slot-minor
    cell_of = np.repeat(np.arange(nC), 6)
    slot_of = np.tile(np.arange(6), nC)
    key = np.sort(loops, axis=1).astype(np.int64)
    nP = int(points.shape[0]) + 1
    # two 64-bit keys are enough for 4 sorted vertex ids < 2^31
    k1 = key[:, 0] * nP + key[:, 1]; k2 = key[:, 2] * nP + key[:, 3]
    order = np.lexsort((np.arange(nC * 6), k2, k1))
    k1s, k2s = k1[order], k2[order]
    newgrp = np.ones(order.size, dtype=bool)
    newgrp[1:] = (k1s[1:] != k1s[:-1]) | (k2s[1:] != k2s[:-1])
    first = order[newgrp]                                      # first (lowest cell) occurrence = owner side
    grp_start = np.nonzero(newgrp)[0]
    grp_size = np.diff(np.append(grp_start, order.size))
    if grp_size.max() > 2:
        raise ValueError("non-manifold face shared by >2 cells")
    second = np.full(first.size, -1, dtype=np.int64)
    two = grp_size == 2
    second[two] = order[grp_start[two] + 1]
    own = cell_of[first]; nei = np.where(second >= 0, cell_of[np.maximum(second, 0)], -1)
    internal = np.nonzero(nei >= 0)[0]; boundary = np.nonzero(nei < 0)[0]
    oi = internal[np.lexsort((nei[internal], own[internal]))]
    patches: List[Tuple[str, str, int, int]] = []
    if classify is not None and patch_names is not None and boundary.size:
        fcen = points[loops[first[boundary]]].mean(1)
        pid = classify(fcen, own[boundary], slot_of[first[boundary]])
        ob_parts = []; pos = oi.size
        for p, (name, typ) in enumerate(patch_names):
            fp = boundary[pid == p]
            fp = fp[np.argsort(own[fp], kind="stable")]
            ob_parts.append(fp); patches.append((name, typ, pos, int(fp.size))); pos += fp.size
        if sum(x.size for x in ob_parts) != boundary.size:
            raise ValueError("boundary faces left unclassified")
        ob = np.concatenate(ob_parts)
    else:
        ob = boundary[np.argsort(own[boundary], kind="stable")]
        patches = [("walls", "wall", int(oi.size), int(ob.size))]
    sel = np.concatenate([oi, ob])
    fv = loops[first[sel]].astype(np.int32).reshape(-1)
    fo = (4 * np.arange(sel.size + 1)).astype(np.int32)
    return PolyMesh(np.ascontiguousarray(points, dtype=np.float64), fo, fv, own[sel].astype(np.int32),
                    nei[oi].astype(np.int32), nC, patches)


def block_mesh(vertices: np.ndarray, blocks: Sequence[Dict], scale: float = 1.0,
               classify=None, patch_names=None, merge_tol: float = 1e-9) -> PolyMesh:
    """``blocks``: dicts with ``hex`` (8 vertex ids, blockMesh order), ``n`` (nx,ny,nz) and either
    ``simple`` (gx,gy,gz) or ``edge`` (12 gradings).  Cells are numbered block by block,
    i fastest then j then k, like blockMesh."""
    vertices = np.asarray(vertices, dtype=np.float64)
    all_pts = []; all_hex = []; base = 0; block_of_cell = []
    for b, blk in enumerate(blocks):
        nx, ny, nz = blk["n"]
        if "edge" in blk:
            eg = list(blk["edge"])
        else:
            gx, gy, gz = blk.get("simple", (1, 1, 1))
            eg = [gx] * 4 + [gy] * 4 + [gz] * 4
        pts = _block_points(vertices[list(blk["hex"])], (nx, ny, nz), eg)
        all_pts.append(pts.reshape(-1, 3))
        pid = (base + np.arange((nx + 1) * (ny + 1) * (nz + 1))).reshape(nz + 1, ny + 1, nx + 1)
        k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        hx = np.stack([pid[k, j, i], pid[k, j, i + 1], pid[k, j + 1, i + 1], pid[k, j + 1, i],
                       pid[k + 1, j, i], pid[k + 1, j, i + 1], pid[k + 1, j + 1, i + 1], pid[k + 1, j + 1, i]], -1)
        all_hex.append(hx.reshape(-1, 8)); block_of_cell.append(np.full(nx * ny * nz, b))
        base += pid.size
    P = np.concatenate(all_pts) * scale
    H = np.concatenate(all_hex)
    # merge coincident block-interface points: union points closer than merge_tol*extent,
    # keep the first-seen one, number survivors in first-seen order (blockMesh-like)
    if len(blocks) > 1:
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import connected_components
        from scipy.spatial import cKDTree
        extent = float(np.linalg.norm(P.max(0) - P.min(0)))
        pairs = cKDTree(P).query_pairs(merge_tol * extent, output_type="ndarray")
        g = coo_matrix((np.ones(len(pairs)), (pairs[:, 0], pairs[:, 1])), shape=(P.shape[0],) * 2)
        _, comp = connected_components(g, directed=False)
        first_idx = np.full(comp.max() + 1, P.shape[0], dtype=np.int64)
        np.minimum.at(first_idx, comp, np.arange(P.shape[0]))
        rank = np.empty(first_idx.size, dtype=np.int64)
        rank[np.argsort(first_idx)] = np.arange(first_idx.size)
        new_id = rank[comp]
        pts = np.empty((first_idx.size, 3)); pts[rank] = P[first_idx]
    else:
        new_id = np.arange(P.shape[0]); pts = P
    boc = np.concatenate(block_of_cell)
    cls = None
    if classify is not None:
        cls = lambda fcen, own, slot: classify(fcen, own, slot, boc[own])  # noqa: E731
    mesh = hexes_to_polymesh(pts, new_id[H], cls, patch_names)
    mesh.block_of_cell = boc  # type: ignore[attr-defined]
    mesh.hexes = new_id[H]    # type: ignore[attr-defined]   (nC, 8) corner ids in blockMesh order (cases/refine.py)
    return mesh


def box_mesh(nx: int, ny: int, nz: int, lower=(0.0, 0.0, 0.0), upper=None, grading=(1, 1, 1)) -> PolyMesh:
    """Single-block hex box; default spans [0,nx]x[0,ny]x[0,nz] like the reference's
    ``HostTetMesh::createBoxMesh`` test geometry (cuda/HostTetMesh.h:62-144)."""
    lo = np.asarray(lower, dtype=np.float64)
    hi = np.asarray(upper if upper is not None else (nx, ny, nz), dtype=np.float64)
    v = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                  [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]])
    return block_mesh(v, [dict(hex=range(8), n=(nx, ny, nz), simple=grading)])
