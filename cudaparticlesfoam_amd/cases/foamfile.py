"""Reading (and writing) a case's mesh and velocity as OpenFOAM stores them on disk -- the data format on the INPUT side of the
path.

The reference's solvers never parse these files themselves: OpenFOAM does, and the fragments take ``mesh.points()``,
``mesh.faces()``, ``mesh.faceOwner()`` / ``faceNeighbour()`` and ``U`` from its objects (``/root/reference/src/initCuda.H:76-124``,
``src/advect.H:44-57``).  The Python host of this repo has no OpenFOAM underneath, so a case directory is brought in here:
``constant/polyMesh/{points, faces, owner, neighbour, boundary}`` and a ``volVectorField`` such as ``<time>/U``, in OpenFOAM's ASCII
format [OpenFOAM, not in the reference tree: FoamFile header, ``N ( ... )`` lists, faces as ``k(v0 ... vk-1)``, ``internalField
uniform (u v w)`` or ``nonuniform List<vector> N ( (u v w) ... )``].  Binary and compressed (.gz) files are refused, not guessed at.
``write_polymesh`` / ``write_vector_field`` produce the same format (tests round-trip through them; handy for handing a synthetic
case of ``cases/`` to a real OpenFOAM installation).
"""
from __future__ import annotations

import os
import re
from typing import List, Optional, Tuple

import numpy as np

from .polymesh import PolyMesh

_HEADER = """/*--------------------------------*- C++ -*----------------------------------*\\
  =========                 |
  \\\\      /  F ield         | written by cudaparticlesfoam_amd.cases.foamfile
   \\\\    /   O peration     |
    \\\\  /    A nd           |
     \\\\/     M anipulation  |
\\*---------------------------------------------------------------------------*/
FoamFile
{
    version     2.0;
    format      ascii;
    class       %s;
%s    location    "%s";
    object      %s;
}
// * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * //

"""


class FoamFormatError(ValueError):
    pass


def _strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _read(path: str) -> Tuple[dict, str]:
    """(FoamFile header entries, body after the header) of an ASCII OpenFOAM file."""
    if path.endswith(".gz") or (not os.path.exists(path) and os.path.exists(path + ".gz")):
        raise FoamFormatError("%s: compressed files are not read (gunzip it first)" % path)
    with open(path, "rb") as f:
        raw = f.read()
    text = _strip_comments(raw.decode("latin-1"))
    m = re.search(r"FoamFile\s*\{(.*?)\}", text, flags=re.S)
    if not m:
        raise FoamFormatError("%s: no FoamFile header" % path)
    head = {}
    for k, v in re.findall(r"(\w+)\s+([^;]*);", m.group(1)):
        head[k] = v.strip().strip('"')
    if head.get("format", "ascii") != "ascii":
        raise FoamFormatError("%s: format %s is not read (foamFormatConvert it to ascii)" % (path, head.get("format")))
    return head, text[m.end():]


def _list_body(body: str, path: str) -> Tuple[int, str]:
    """The first ``N ( ... )`` list of the body: (N, text between the outer parentheses)."""
    m = re.search(r"(\d+)\s*\(", body)
    if not m:
        raise FoamFormatError("%s: no list found" % path)
    n = int(m.group(1))
    depth, i = 1, m.end()
    while i < len(body) and depth:
        c = body[i]
        depth += (c == "(") - (c == ")")
        i += 1
    if depth:
        raise FoamFormatError("%s: unbalanced parentheses" % path)
    return n, body[m.end():i - 1]


def _numbers(text: str, dtype):
    return np.array(re.sub(r"[()]", " ", text).split(), dtype=dtype)


def read_labels(path: str) -> Tuple[np.ndarray, dict]:
    head, body = _read(path)
    n, inner = _list_body(body, path)
    a = _numbers(inner, np.int64)
    if a.size != n:
        raise FoamFormatError("%s: %d labels announced, %d found" % (path, n, a.size))
    return a, head


def read_points(path: str) -> np.ndarray:
    _, body = _read(path)
    n, inner = _list_body(body, path)
    a = _numbers(inner, np.float64)
    if a.size != 3 * n:
        raise FoamFormatError("%s: %d points announced, %d numbers found" % (path, n, a.size))
    return a.reshape(n, 3)


def read_faces(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """(face_offsets [nFaces + 1], face_verts) from a ``faceList``: entries ``k(v0 ... vk-1)``."""
    _, body = _read(path)
    n, inner = _list_body(body, path)
    sizes, verts = [], []
    for k, vs in re.findall(r"(\d+)\s*\(([^()]*)\)", inner):
        v = vs.split()
        if len(v) != int(k):
            raise FoamFormatError("%s: a face announces %s vertices and lists %d" % (path, k, len(v)))
        sizes.append(int(k)); verts.extend(v)
    if len(sizes) != n:
        raise FoamFormatError("%s: %d faces announced, %d found" % (path, n, len(sizes)))
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    return off, np.array(verts, dtype=np.int64)


def read_boundary(path: str) -> List[Tuple[str, str, int, int]]:
    """[(patch name, type, startFace, nFaces)] of a ``polyBoundaryMesh`` file; [] if the file does not exist."""
    if not os.path.exists(path):
        return []
    _, body = _read(path)
    _, inner = _list_body(body, path)
    out = []
    for name, block in re.findall(r"(\w+)\s*\{(.*?)\}", inner, flags=re.S):
        ent = dict(re.findall(r"(\w+)\s+([^;]*);", block))
        out.append((name, ent.get("type", "patch").strip(), int(ent["startFace"]), int(ent["nFaces"])))
    return out


def read_polymesh(case_dir: str, region: Optional[str] = None) -> PolyMesh:
    """``<case>/constant[/<region>]/polyMesh`` -> PolyMesh (32-bit labels where they fit, like ``cpf_set_mesh``; 64-bit otherwise:
    ``cpf_set_mesh_l64``)."""
    d = os.path.join(case_dir, "constant", region or "", "polyMesh")
    points = read_points(os.path.join(d, "points"))
    off, verts = read_faces(os.path.join(d, "faces"))
    owner, head = read_labels(os.path.join(d, "owner"))
    neighbour, _ = read_labels(os.path.join(d, "neighbour"))
    if owner.size != off.size - 1 or neighbour.size > owner.size:
        raise FoamFormatError("%s: %d faces, %d owners, %d neighbours" % (d, off.size - 1, owner.size, neighbour.size))
    m = re.search(r"nCells:\s*(\d+)", head.get("note", ""))
    n_cells = int(m.group(1)) if m else int(max(owner.max(), neighbour.max() if neighbour.size else 0)) + 1
    if verts.size and (verts.min() < 0 or verts.max() >= points.shape[0]):
        raise FoamFormatError("%s: face vertex out of range" % d)
    lab = np.int32 if max(points.shape[0], verts.size, n_cells) < 2 ** 31 - 16 else np.int64
    return PolyMesh(points=points, face_offsets=off.astype(lab), face_verts=verts.astype(lab), owner=owner.astype(lab),
                    neighbour=neighbour.astype(lab), n_cells=n_cells, patches=read_boundary(os.path.join(d, "boundary")))


def read_vector_field(path: str, n_cells: int) -> np.ndarray:
    """The ``internalField`` of a volVectorField (``0/U``): (n_cells, 3) float64."""
    _, body = _read(path)
    m = re.search(r"internalField\s+(uniform|nonuniform)\s*", body)
    if not m:
        raise FoamFormatError("%s: no internalField" % path)
    rest = body[m.end():]
    if m.group(1) == "uniform":
        mm = re.match(r"\(\s*([^()]*)\)", rest)
        if not mm:
            raise FoamFormatError("%s: uniform value is not a vector" % path)
        v = np.array(mm.group(1).split(), dtype=np.float64)
        if v.size != 3:
            raise FoamFormatError("%s: uniform value is not a vector" % path)
        return np.tile(v, (n_cells, 1))
    if not re.match(r"List<vector>", rest):
        raise FoamFormatError("%s: internalField is not a List<vector>" % path)
    n, inner = _list_body(rest, path)
    a = _numbers(inner, np.float64)
    if n != n_cells or a.size != 3 * n:
        raise FoamFormatError("%s: %d cells expected, List<vector> of %d with %d numbers" % (path, n_cells, n, a.size))
    return a.reshape(n, 3)


# ------------------------------------------------------------------------------------------------ writing
def _open(path: str, cls: str, obj: str, location: str, note: str = ""):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    f = open(path, "w")
    f.write(_HEADER % (cls, ('    note        "%s";\n' % note) if note else "", location, obj))
    return f


def write_polymesh(mesh: PolyMesh, case_dir: str, region: Optional[str] = None) -> str:
    """Writes ``constant[/<region>]/polyMesh/{points, faces, owner, neighbour, boundary}`` (ASCII, 17 significant digits: the
    doubles survive the round trip bit for bit).  Returns the polyMesh directory."""
    loc = os.path.join("constant", region or "", "polyMesh").replace(os.sep + os.sep, os.sep)
    d = os.path.join(case_dir, loc)
    note = "nPoints:%d  nCells:%d  nFaces:%d  nInternalFaces:%d" % (mesh.n_points, mesh.n_cells, mesh.n_faces, mesh.n_internal)
    with _open(os.path.join(d, "points"), "vectorField", "points", loc) as f:
        f.write("%d\n(\n" % mesh.n_points)
        f.write("\n".join("(%.17g %.17g %.17g)" % tuple(p) for p in np.asarray(mesh.points, dtype=np.float64)))
        f.write("\n)\n")
    fo = np.asarray(mesh.face_offsets, dtype=np.int64); fv = np.asarray(mesh.face_verts, dtype=np.int64)
    with _open(os.path.join(d, "faces"), "faceList", "faces", loc) as f:
        f.write("%d\n(\n" % mesh.n_faces)
        f.write("\n".join("%d(%s)" % (fo[k + 1] - fo[k], " ".join(map(str, fv[fo[k]:fo[k + 1]]))) for k in range(mesh.n_faces)))
        f.write("\n)\n")
    for name, arr in (("owner", mesh.owner), ("neighbour", mesh.neighbour)):
        with _open(os.path.join(d, name), "labelList", name, loc, note) as f:
            a = np.asarray(arr, dtype=np.int64)
            f.write("%d\n(\n%s\n)\n" % (a.size, "\n".join(map(str, a))))
    patches = list(mesh.patches) or [("defaultFaces", "patch", mesh.n_internal, mesh.n_faces - mesh.n_internal)]
    with _open(os.path.join(d, "boundary"), "polyBoundaryMesh", "boundary", loc) as f:
        f.write("%d\n(\n" % len(patches))
        for name, typ, start, size in patches:
            f.write("    %s\n    {\n        type            %s;\n        nFaces          %d;\n        startFace       %d;\n    }\n"
                    % (name, typ, size, start))
        f.write(")\n")
    return d


def write_vector_field(U: np.ndarray, path: str, name: str = "U", dimensions: str = "[0 1 -1 0 0 0 0]") -> None:
    """A volVectorField with a nonuniform internalField and ``zeroGradient``-free, empty boundaryField (enough for this repo's
    reader and for ``foamDictionary``; a solver wants the case's own boundary conditions)."""
    U = np.asarray(U, dtype=np.float64)
    with _open(path, "volVectorField", name, os.path.basename(os.path.dirname(path))) as f:
        f.write("dimensions      %s;\n\ninternalField   nonuniform List<vector>\n%d\n(\n" % (dimensions, U.shape[0]))
        f.write("\n".join("(%.17g %.17g %.17g)" % tuple(u) for u in U))
        f.write("\n)\n;\n\nboundaryField\n{\n}\n")
