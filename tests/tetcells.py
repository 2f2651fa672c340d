"""Test helper: a polyMesh whose CELLS are the tets of the reference's own test geometry
(HostTetMesh::createBoxMesh, cuda/HostTetMesh.h:62-144: 6 tets per unit cube).  On such a mesh the
product's polyhedral-cell walk and the reference's tet walk operate on the very same elements, so they can
be compared without any decomposition in between; it also exercises the generic (non-hex) CSR path."""
import numpy as np

from cudaparticlesfoam_amd.cases import build_polymesh_from_cells


def box_tets(nx, ny, nz):
    """(positions, tets) exactly as createBoxMesh lays them out (restated; checked against oracle/_ref in tests)."""
    xs, ys, zs = np.arange(nx + 1.0), np.arange(ny + 1.0), np.arange(nz + 1.0)
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    pos = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    tets = []
    for iz in range(nz):
        for iy in range(ny):
            for ix in range(nx):
                v0 = iz * (nx + 1) * (ny + 1) + iy * (nx + 1) + ix
                v1 = v0 + 1; v2 = v0 + (nx + 1); v3 = v1 + (nx + 1)
                v4 = v0 + (nx + 1) * (ny + 1); v5 = v1 + (nx + 1) * (ny + 1)
                v6 = v2 + (nx + 1) * (ny + 1); v7 = v3 + (nx + 1) * (ny + 1)
                tets += [(v0, v1, v3, v7), (v0, v1, v7, v5), (v0, v5, v7, v4), (v0, v3, v2, v7), (v0, v6, v4, v7),
                         (v0, v2, v6, v7)]
    return pos, np.asarray(tets, dtype=np.int32)


def tet_cell_polymesh(pos, tets):
    """polyMesh with one 4-faced cell per tet; cell id == tet id."""
    loops = []
    for t in tets:
        a, b, c, d = (int(v) for v in t)
        vol = np.dot(pos[d] - pos[a], np.cross(pos[b] - pos[a], pos[c] - pos[a]))
        if vol < 0:
            a, b = b, a
        # faces opposite each vertex, counter-clockwise seen from outside for a positively oriented tet
        loops.append([(b, c, d), (a, d, c), (a, b, d), (a, c, b)])
    return build_polymesh_from_cells(pos, loops)
