"""Extruded polygon meshes: test meshes with truly polyhedral cells (more than six distinct face planes).

``cut_corner_box``: an nx x ny grid of unit squares with the upper-right corner of every ``every``-th square cut off, then
extruded nz layers in z.  A cut square becomes a pentagonal prism (SEVEN distinct planes: header record + CSR walk in the
streaming kernel), the corner a triangular prism (five: padded record), and the two squares that share the cut edges get a
hanging node -- their side face there is two coplanar pieces leading to different cells (a face group; cpf_mesh.cpp).
The reference cannot run such meshes (src/initCuda.H:64); they exercise this repo's polyhedral paths.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .polymesh import PolyMesh, build_polymesh_from_cells


def extrude_polygons(points2d: np.ndarray, loops: List[List[int]], zs) -> PolyMesh:
    """2-D cells (counter-clockwise vertex loops; collinear hanging nodes allowed) -> prisms, one layer per z interval."""
    zs = np.asarray(zs, dtype=np.float64)
    n2 = points2d.shape[0]
    pts = np.concatenate([np.column_stack([points2d, np.full(n2, z)]) for z in zs])
    cells = []
    for k in range(zs.size - 1):
        lo, hi = k * n2, (k + 1) * n2
        for loop in loops:
            faces = [tuple(lo + v for v in reversed(loop)), tuple(hi + v for v in loop)]
            for a, b in zip(loop, loop[1:] + loop[:1]):
                faces.append((lo + a, lo + b, hi + b, hi + a))
            cells.append(faces)
    return build_polymesh_from_cells(pts, cells)


def cut_corner_box(nx: int, ny: int, nz: int, every: int = 3, cut: float = 0.4) -> Tuple[PolyMesh, Dict[str, int]]:
    """See the module docstring.  Returns the mesh over [0, nx] x [0, ny] x [0, nz] and the number of 2-D cells of each kind."""
    vid = lambda i, j: j * (nx + 1) + i                                    # noqa: E731
    pts = [(float(i), float(j)) for j in range(ny + 1) for i in range(nx + 1)]
    on_vertical: Dict[Tuple[int, int], int] = {}                           # hanging node on the edge (i, j) -> (i, j + 1)
    on_horizontal: Dict[Tuple[int, int], int] = {}                         # ... on the edge (i, j) -> (i + 1, j)
    cuts = [(i, j) for j in range(ny) for i in range(nx) if (i + 2 * j) % every == 0]
    for i, j in cuts:                                                      # the corner at (i + 1, j + 1)
        on_vertical[(i + 1, j)] = len(pts); pts.append((i + 1.0, j + 1.0 - cut))
        on_horizontal[(i, j + 1)] = len(pts); pts.append((i + 1.0 - cut, j + 1.0))
    loops: List[List[int]] = []
    kinds = {"pentagons": 0, "triangles": 0, "squares_with_hanging_node": 0, "squares": 0}
    cutset = set(cuts)
    for j in range(ny):
        for i in range(nx):
            if (i, j) in cutset:
                m1, m2 = on_vertical[(i + 1, j)], on_horizontal[(i, j + 1)]
                loop = [vid(i, j)]
                if (i, j) in on_horizontal: loop.append(on_horizontal[(i, j)])          # noqa: E701
                loop += [vid(i + 1, j), m1, m2, vid(i, j + 1)]
                if (i, j) in on_vertical: loop.append(on_vertical[(i, j)])              # noqa: E701
                loops.append(loop); kinds["pentagons"] += 1
                loops.append([m1, vid(i + 1, j + 1), m2]); kinds["triangles"] += 1
                continue
            loop = [vid(i, j)]                                             # bottom, right, top, left edges, counter-clockwise
            if (i, j) in on_horizontal: loop.append(on_horizontal[(i, j)])              # noqa: E701
            loop.append(vid(i + 1, j))
            loop.append(vid(i + 1, j + 1))
            loop.append(vid(i, j + 1))
            if (i, j) in on_vertical: loop.append(on_vertical[(i, j)])                  # noqa: E701
            kinds["squares_with_hanging_node" if len(loop) > 4 else "squares"] += 1
            loops.append(loop)
    return extrude_polygons(np.asarray(pts), loops, np.arange(nz + 1, dtype=np.float64)), kinds
