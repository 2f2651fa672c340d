"""2:1 local refinement of a hex mesh -> a genuinely POLYHEDRAL polyMesh (synthetic test / bench input).

What ``snappyHexMesh`` / ``refineMesh`` leave at a refinement interface [OpenFOAM, not in the reference tree]: the
refined hexes become 8 hexes (4 when the mesh is only refined in x and y, the way a one-cell-thick 2-D case is), and
an UNREFINED cell next to a refined one keeps its shape but has the shared face replaced by the 4 (2) faces of the
refined side -- a cell with 9 (7) faces, up to 24 (10) when all its neighbours are refined.  The reference cannot run
such meshes at all (``src/initCuda.H:64``: ``tetsPerCell = 12``); BASELINE.json ``configs[4]`` names them
("motorBike-scale polyMesh"), so the walk and its fast path are tested on them.

Sub-points are the averages of the parent corners they sit between (edge mid-points, face centres, cell centre), so
the pieces of a planar face lie in the parent face's plane and a hanging node lies on the straight edge it splits.
Faces of unrefined cells that merely TOUCH a split edge keep their four vertices (same plane either way).
"""
from __future__ import annotations

from typing import Dict, FrozenSet, List, Tuple

import numpy as np

from .blockmesh import HEX_FACES
from .polymesh import PolyMesh, build_polymesh_from_cells

# blockMesh corner order: local index of the corner at (x-bit, y-bit, z-bit)
_CORNER = {(0, 0, 0): 0, (1, 0, 0): 1, (1, 1, 0): 2, (0, 1, 0): 3, (0, 0, 1): 4, (1, 0, 1): 5, (1, 1, 1): 6, (0, 1, 1): 7}
_ORDER = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]


def refine_hexes(points: np.ndarray, hexes: np.ndarray, mask: np.ndarray, split_z: bool = True):
    """``hexes`` (nC, 8) in blockMesh corner order, ``mask`` (nC,) bool: cells to split 2 x 2 x 2 (``split_z``) or
    2 x 2 x 1.  Returns (PolyMesh, parent) with ``parent[c]`` = the input cell that output cell ``c`` is (a piece of).
    Output cells keep the input order, the pieces of a refined cell taking its place."""
    points = np.asarray(points, dtype=np.float64)
    hexes = np.asarray(hexes, dtype=np.int64)
    mask = np.asarray(mask, dtype=bool)
    pts: List[np.ndarray] = [p for p in points]
    node_of: Dict[FrozenSet[int], int] = {}

    def node(ids) -> int:
        key = frozenset(int(i) for i in ids)
        if len(key) == 1:
            return next(iter(key))
        n = node_of.get(key)
        if n is None:
            n = len(pts)
            node_of[key] = n
            pts.append(points[sorted(key)].mean(0))
        return n

    bits = {0: (0,), 1: (0, 1), 2: (1,)}                      # lattice coordinate -> parent corner bits involved
    split_faces = set()
    for c in np.nonzero(mask)[0]:
        h = hexes[c]
        for f in HEX_FACES:
            split_faces.add(tuple(sorted(int(v) for v in h[f])))
    # pass 1: the pieces of the refined cells (creates every sub-point)
    pieces: Dict[int, List[List[Tuple[int, ...]]]] = {}
    for c in np.nonzero(mask)[0]:
        h = hexes[c]

        def lat(a, b, cc):
            return node(h[_CORNER[(xb, yb, zb)]] for xb in bits[a] for yb in bits[b] for zb in bits[cc])
        out = []
        for k in ((0, 1) if split_z else (0,)):
            for j in (0, 1):
                for i in (0, 1):
                    ch = [lat(i + dx, j + dy, (k + dz) if split_z else 2 * dz) for dx, dy, dz in _ORDER]
                    out.append([tuple(ch[v] for v in f) for f in HEX_FACES])
        pieces[int(c)] = out
    # pass 2: cells in input order; unrefined cells take over the sub-faces of their refined neighbours
    cells: List[List[Tuple[int, ...]]] = []
    parent: List[int] = []
    for c in range(hexes.shape[0]):
        h = hexes[c]
        if mask[c]:
            for piece in pieces[c]:
                cells.append(piece)
                parent.append(c)
            continue
        loops: List[Tuple[int, ...]] = []
        for f in HEX_FACES:
            q = [int(v) for v in h[f]]
            if tuple(sorted(q)) not in split_faces:
                loops.append(tuple(q))
                continue
            mid = [node_of.get(frozenset((q[i], q[(i + 1) % 4]))) for i in range(4)]
            ctr = node_of.get(frozenset(q))
            if ctr is not None:                                # split in four: the refined side was cut both ways here
                for i in range(4):
                    loops.append((q[i], mid[i], ctr, mid[(i + 3) % 4]))
            else:                                              # split in two (2 x 2 x 1 refinement, a face with a z edge)
                i = 0 if (mid[0] is not None and mid[2] is not None) else 1
                assert mid[i] is not None and mid[i + 2] is not None, "face of a refined neighbour without split edges"
                loops.append((q[i], mid[i], mid[i + 2], q[(i + 3) % 4]))
                loops.append((mid[i], q[(i + 1) % 4], q[(i + 2) % 4], mid[i + 2]))
        cells.append(loops)
        parent.append(c)
    mesh = build_polymesh_from_cells(np.asarray(pts), cells)
    return mesh, np.asarray(parent, dtype=np.int64)


def refined_box(nx: int, ny: int, nz: int, lower, upper, region, grading=(1, 1, 1)):
    """A graded box of hexes whose cells with centres inside ``region`` = ((x0,y0,z0),(x1,y1,z1)) are split 2x2x2."""
    from .blockmesh import box_mesh
    m = box_mesh(nx, ny, nz, lower=lower, upper=upper, grading=grading)
    c, _ = m.cell_centres_volumes()
    lo, hi = np.asarray(region[0]), np.asarray(region[1])
    mask = np.all((c >= lo) & (c <= hi), axis=1)
    return refine_hexes(m.points, m.hexes, mask, split_z=True)


def refined_pitzdaily(region=((0.0, -0.0254), (0.06, 0.0254))):
    """pitzDaily (12 225 hexes, one cell thick) with the cells whose centre lies in the (x, y) ``region`` -- by default
    the first 60 mm behind the step, where the tutorial's recirculation sits -- split 2 x 2 x 1: the unrefined cells
    along the patch's rim have 7 (at a corner of the patch 8) faces."""
    from . import pitzdaily as pz
    m = pz.pitzdaily_mesh()
    c, _ = m.cell_centres_volumes()
    (x0, y0), (x1, y1) = region
    mask = (c[:, 0] >= x0) & (c[:, 0] <= x1) & (c[:, 1] >= y0) & (c[:, 1] <= y1)
    return refine_hexes(m.points, m.hexes, mask, split_z=False)
