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


def chamfered_box(nx: int, ny: int, nz: int, cuts_per_corner: int = 1, period: int = 3, cut: float = 0.3) -> Tuple[PolyMesh, Dict[str, int]]:
    """An nx x ny grid of unit squares in which every square with i % period == 1 and j % period == 1 has ALL FOUR corners
    chamfered -- with one cut an octagon (prism: TEN distinct planes: a two-record cell, cpf_walk.h), with two cuts per
    corner a dodecagon (FOURTEEN: beyond two records, header record + CSR walk) -- extruded nz layers in z.  The corner
    pieces are triangular prisms (five planes); the four edge neighbours of a chamfered square get two hanging nodes on the
    shared edge, i.e. a face group of three coplanar pieces.  Returns the mesh over [0, nx] x [0, ny] x [0, nz] and the number
    of 2-D cells of each kind."""
    assert cuts_per_corner in (1, 2) and period >= 3 and 0.0 < cut < 0.5
    vid = lambda i, j: j * (nx + 1) + i                                    # noqa: E731
    pts: List[Tuple[float, float]] = [(float(i), float(j)) for j in range(ny + 1) for i in range(nx + 1)]
    hanging_h: Dict[Tuple[int, int], List[Tuple[float, int]]] = {}         # on the edge (i, j) -> (i + 1, j): (x, vertex)
    hanging_v: Dict[Tuple[int, int], List[Tuple[float, int]]] = {}         # on the edge (i, j) -> (i, j + 1): (y, vertex)

    def new_point(x, y):
        pts.append((x, y)); return len(pts) - 1

    chosen = [(i, j) for j in range(ny) for i in range(nx) if i % period == 1 and j % period == 1]
    chosen_set = set(chosen)
    loops: List[List[int]] = []
    kinds = {"polygons": 0, "triangles": 0, "squares_with_hanging_nodes": 0, "squares": 0}
    e = cut / 6.0
    for i, j in chosen:
        x0, y0, x1, y1 = float(i), float(j), i + 1.0, j + 1.0
        # chamfer end points on the four edges, in the polygon's counter-clockwise order
        pL0, pB0 = new_point(x0, y0 + cut), new_point(x0 + cut, y0)        # bottom-left corner
        pB1, pR0 = new_point(x1 - cut, y0), new_point(x1, y0 + cut)        # bottom-right
        pR1, pT1 = new_point(x1, y1 - cut), new_point(x1 - cut, y1)        # top-right
        pT0, pL1 = new_point(x0 + cut, y1), new_point(x0, y1 - cut)        # top-left
        hanging_h[(i, j)] = [(x0 + cut, pB0), (x1 - cut, pB1)]
        hanging_h[(i, j + 1)] = [(x0 + cut, pT0), (x1 - cut, pT1)]
        hanging_v[(i, j)] = [(y0 + cut, pL0), (y1 - cut, pL1)]
        hanging_v[(i + 1, j)] = [(y0 + cut, pR0), (y1 - cut, pR1)]
        corners = [(vid(i, j), pL0, pB0, (x0 + cut / 2 - e, y0 + cut / 2 - e)),
                   (vid(i + 1, j), pB1, pR0, (x1 - cut / 2 + e, y0 + cut / 2 - e)),
                   (vid(i + 1, j + 1), pR1, pT1, (x1 - cut / 2 + e, y1 - cut / 2 + e)),
                   (vid(i, j + 1), pT0, pL1, (x0 + cut / 2 - e, y1 - cut / 2 + e))]
        poly: List[int] = []
        for V, first, last, mid in corners:
            if cuts_per_corner == 1:
                poly += [first, last]
                loops.append([V, last, first]); kinds["triangles"] += 1
            else:
                m = new_point(*mid)
                poly += [first, m, last]
                loops.append([V, m, first]); loops.append([V, last, m]); kinds["triangles"] += 2
        loops.append(poly); kinds["polygons"] += 1
    for j in range(ny):
        for i in range(nx):
            if (i, j) in chosen_set:
                continue
            loop = [vid(i, j)] + [v for _, v in sorted(hanging_h.get((i, j), []))]                     # bottom edge, left to right
            loop += [vid(i + 1, j)] + [v for _, v in sorted(hanging_v.get((i + 1, j), []))]            # right edge, upwards
            loop += [vid(i + 1, j + 1)] + [v for _, v in sorted(hanging_h.get((i, j + 1), []), reverse=True)]   # top edge, right to left
            loop += [vid(i, j + 1)] + [v for _, v in sorted(hanging_v.get((i, j), []), reverse=True)]  # left edge, downwards
            kinds["squares_with_hanging_nodes" if len(loop) > 4 else "squares"] += 1
            loops.append(loop)
    return extrude_polygons(np.asarray(pts), loops, np.arange(nz + 1, dtype=np.float64)), kinds


def diamond_box(nx: int, ny: int, nz: int, period: int = 6, cut: float = 0.3) -> Tuple[PolyMesh, Dict[str, int]]:
    """A CONFORMAL mesh with a minority of true polyhedra and no hanging nodes (what a surface-snapped hex-dominant mesh
    looks like away from its refinement interfaces): an nx x ny grid of unit squares in which, at every interior grid vertex
    with i % period == 0 and j % period == 0, the four squares around the vertex lose that corner and a diamond (a square on
    its tip) fills the hole.  The four squares become pentagonal prisms (SEVEN planes: two-record cells), the diamond is a
    six-plane cell with slanted side faces; period 6: 11 % of the cells have seven planes, period 1: every square loses all
    four corners (octagonal prisms, TEN planes, half of the cells).  Extruded nz layers in z."""
    assert period >= 1 and 0.0 < cut < 0.5
    vid = lambda i, j: j * (nx + 1) + i                                    # noqa: E731
    pts: List[Tuple[float, float]] = [(float(i), float(j)) for j in range(ny + 1) for i in range(nx + 1)]
    chosen = {(i, j) for j in range(1, ny) for i in range(1, nx) if i % period == 0 and j % period == 0}
    arms: Dict[Tuple[int, int], Tuple[int, int, int, int]] = {}            # vertex -> its diamond's points towards +x, +y, -x, -y
    for i, j in sorted(chosen):
        e = []
        for dx, dy in ((cut, 0.0), (0.0, cut), (-cut, 0.0), (0.0, -cut)):
            pts.append((i + dx, j + dy)); e.append(len(pts) - 1)
        arms[(i, j)] = tuple(e)
    loops: List[List[int]] = []
    kinds = {"diamonds": 0, "polygons": 0, "squares": 0}
    for (i, j), (px, py, mx, my) in sorted(arms.items()):
        loops.append([px, py, mx, my]); kinds["diamonds"] += 1
    for j in range(ny):
        for i in range(nx):
            loop: List[int] = []
            # counter-clockwise: bottom-left, bottom-right, top-right, top-left corner; a chosen corner is replaced by the two
            # diamond points on this square's edges (in traversal order)
            for (ci, cj), (first, last) in (((i, j), (1, 0)), ((i + 1, j), (2, 1)), ((i + 1, j + 1), (3, 2)), ((i, j + 1), (0, 3))):
                if (ci, cj) in arms:
                    a = arms[(ci, cj)]
                    loop += [a[first], a[last]]
                else:
                    loop.append(vid(ci, cj))
            kinds["polygons" if len(loop) > 4 else "squares"] += 1
            loops.append(loop)
    return extrude_polygons(np.asarray(pts), loops, np.arange(nz + 1, dtype=np.float64)), kinds
