"""CPU: the case directory as OpenFOAM stores it (cases/foamfile.py) -- the data format on the input side of the path.  The
reference's solvers get mesh and U from OpenFOAM objects (src/initCuda.H:76-124, src/advect.H:44-57); the Python host reads the
files: round trips of this repo's synthetic cases bit for bit, hand-written files in OpenFOAM's ASCII style, refusals, and (where
the reference tree is present) the tutorials' own 0/U."""
import os

import numpy as np
import pytest

from cudaparticlesfoam_amd.cases import foamfile as ff


def _same_mesh(a, b):
    assert a.n_cells == b.n_cells and np.array_equal(a.points, b.points)             # %.17g: doubles survive bit for bit
    assert np.array_equal(a.face_offsets, b.face_offsets) and np.array_equal(a.face_verts, b.face_verts)
    assert np.array_equal(a.owner, b.owner) and np.array_equal(a.neighbour, b.neighbour)


def test_polymesh_round_trip_pitzdaily_and_polyhedra(tmp_path, pitz):
    from cudaparticlesfoam_amd.cases.polygons import chamfered_box
    for k, mesh in enumerate((pitz["mesh"], chamfered_box(6, 6, 2, 2)[0])):
        case = tmp_path / ("case%d" % k)
        d = ff.write_polymesh(mesh, str(case))
        assert sorted(os.listdir(d)) == ["boundary", "faces", "neighbour", "owner", "points"]
        back = ff.read_polymesh(str(case))
        _same_mesh(mesh, back)
        assert back.face_verts.dtype == np.int32
        assert sum(p[3] for p in back.patches) == mesh.n_faces - mesh.n_internal
    # pitzDaily keeps its five patches (inlet, outlet, upperWall, lowerWall, frontAndBack) with their ranges
    back = ff.read_polymesh(str(tmp_path / "case0"))
    assert [p[0] for p in back.patches] == [p[0] for p in pitz["mesh"].patches] and back.patches == list(pitz["mesh"].patches)


def test_tables_built_from_the_files_equal_the_tables_built_from_memory(tmp_path, pitz):
    """What matters downstream: cpf_set_mesh sees the same arrays, so the walk tables are bit-identical."""
    from cudaparticlesfoam_amd.api import build_mesh_tables_host
    ff.write_polymesh(pitz["mesh"], str(tmp_path / "c"))
    a, b = build_mesh_tables_host(pitz["mesh"]), build_mesh_tables_host(ff.read_polymesh(str(tmp_path / "c")))
    assert all(np.array_equal(a[k], b[k]) for k in a)


def test_vector_field_round_trip_and_uniform(tmp_path, pitz):
    U = pitz["U_analytic"]
    p = tmp_path / "c" / "0" / "U"
    ff.write_vector_field(U, str(p))
    assert np.array_equal(ff.read_vector_field(str(p), U.shape[0]), U)
    with pytest.raises(ff.FoamFormatError):
        ff.read_vector_field(str(p), U.shape[0] + 1)
    q = tmp_path / "c" / "0" / "Uuni"
    q.write_text("""FoamFile { version 2.0; format ascii; class volVectorField; object U; }
dimensions [0 1 -1 0 0 0 0];   // a comment
internalField   uniform (10 0 -2.5e-1);
boundaryField { inlet { type fixedValue; value uniform (1 2 3); } }
""")
    u = ff.read_vector_field(str(q), 7)
    assert u.shape == (7, 3) and (u == [10.0, 0.0, -0.25]).all()


def test_hand_written_files_in_openfoam_style(tmp_path):
    """Two hexes sharing a face, written the way OpenFOAM writes them: comments, the `note` entry with nCells, inGroups lists."""
    d = tmp_path / "two" / "constant" / "polyMesh"
    d.mkdir(parents=True)
    head = "FoamFile\n{\n    version 2.0;\n    format ascii;\n    class %s;\n%s    location \"constant/polyMesh\";\n    object %s;\n}\n"
    (d / "points").write_text(head % ("vectorField", "", "points") + """
12
(
(0 0 0) (1 0 0) (2 0 0)     /* three per line */
(0 1 0) (1 1 0) (2 1 0)
(0 0 1) (1 0 1) (2 0 1)
(0 1 1) (1 1 1) (2 1 1)
)
""")
    (d / "faces").write_text(head % ("faceList", "", "faces") + """
11
(
4(1 4 10 7)       // the shared face
4(0 6 9 3)
4(2 5 11 8)
4(0 1 7 6)
4(1 2 8 7)
4(3 9 10 4)
4(4 10 11 5)
4(0 3 4 1)
4(1 4 5 2)
4(6 7 10 9)
4(7 8 11 10)
)
""")
    note = '    note        "nPoints:12  nCells:2  nFaces:11  nInternalFaces:1";\n'
    (d / "owner").write_text(head % ("labelList", note, "owner") + "\n11\n(\n0 0 1 0 1 0 1 0 1 0 1\n)\n")
    (d / "neighbour").write_text(head % ("labelList", note, "neighbour") + "\n1\n(\n1\n)\n")
    (d / "boundary").write_text(head % ("polyBoundaryMesh", "", "boundary") + """
2
(
    sides
    {
        type            wall;
        inGroups        1(wall);
        nFaces          6;
        startFace       1;
    }
    frontAndBack
    {
        type            empty;
        nFaces          4;
        startFace       7;
    }
)
""")
    m = ff.read_polymesh(str(tmp_path / "two"))
    assert (m.n_cells, m.n_points, m.n_faces, m.n_internal) == (2, 12, 11, 1)
    assert m.patches == [("sides", "wall", 1, 6), ("frontAndBack", "empty", 7, 4)]
    _, vol = m.cell_centres_volumes()
    assert np.allclose(vol, [1.0, 1.0])
    from cudaparticlesfoam_amd.api import mesh_flags_host
    assert mesh_flags_host(m)["all_hex"] == 1


def test_binary_and_compressed_files_are_refused(tmp_path):
    d = tmp_path / "b" / "constant" / "polyMesh"
    d.mkdir(parents=True)
    (d / "points").write_text("FoamFile { version 2.0; format binary; class vectorField; object points; }\n3\n(xxxx)\n")
    with pytest.raises(ff.FoamFormatError, match="binary"):
        ff.read_points(str(d / "points"))
    (d / "owner.gz").write_bytes(b"\x1f\x8b")
    with pytest.raises(ff.FoamFormatError, match="compressed"):
        ff.read_labels(str(d / "owner"))
    (d / "faces").write_text("FoamFile { format ascii; class faceList; object faces; }\n2\n(\n3(0 1 2)\n4(0 1 2)\n)\n")
    with pytest.raises(ff.FoamFormatError, match="announces"):
        ff.read_faces(str(d / "faces"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/tutorials"), reason="the reference tree is only present in the build container")
def test_the_tutorials_own_U_files_parse():
    for case, n in (("cudaParticlesUncoupledFoam/pitzDaily", 12225), ("cudaParticlesPimpleFoam/TJunction", 248000)):
        u = ff.read_vector_field(os.path.join("/root/reference/tutorials/incompressible", case, "0", "U"), n)
        assert u.shape == (n, 3) and (u == 0.0).all()              # both start from internalField uniform (0 0 0)
