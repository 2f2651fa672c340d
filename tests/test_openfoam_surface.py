"""CPU: the OpenFOAM surface of the replacement fragments is exactly what INTEGRATION.md section 6 lists.

The fragments (compat/src/initCuda.H, advect.H) have only ever been compiled against the mock (compat/mock_openfoam/fvCFD.H):
the drop-in claim rests on the mock's fidelity.  The table turns "compiles against our mock" into a claim a maintainer can diff
against OpenFOAM v2106; this test fails when a fragment starts using a mock member that has no row, when a row is not used (or
not declared by the mock) any more, and when one of the solver's objects is used through a member that is not listed."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "cudaparticlesfoam_amd", "compat")

CPP_WORDS = set("""int double const static inline struct template class typename return if else for while void bool true false using
namespace typedef unsigned long char size_t auto this operator explicit public private sizeof std string vector map nullptr
static_cast reinterpret_cast const_cast thread_local define include pragma once mutex condition_variable unique_lock cout cerr
endl ostream ostringstream exit data T n v d f s i r k g it a b c t l e""".split())


def _code(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//.*", "", text)
    return re.sub(r'"(\\.|[^"\\])*"', '""', text)


def _ids(text):
    return set(re.findall(r"[A-Za-z_]\w*", _code(text)))


def _read(*parts):
    return open(os.path.join(COMPAT, *parts)).read()


def _table():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    body = text.split("<!-- openfoam-surface:begin -->")[1].split("<!-- openfoam-surface:end -->")[0]
    rows = {}
    for line in body.strip().splitlines()[2:]:
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        assert len(cells) == 3 and cells[1] and cells[2], line
        for name in re.findall(r"`([^`]+)`", cells[0]):
            assert name not in rows, "two rows for " + name
            rows[name] = (cells[1], cells[2])
    return rows


def test_every_mock_member_a_fragment_uses_has_a_row_and_every_row_is_used():
    mock = _ids(_read("mock_openfoam", "fvCFD.H"))
    frag = _ids(_read("src", "initCuda.H")) | _ids(_read("src", "advect.H"))
    used = (frag & mock) - CPP_WORDS
    table = _table()
    assert used - set(table) == set(), "used by a fragment, declared by the mock, missing from INTEGRATION.md section 6: %s" % sorted(used - set(table))
    assert set(table) - used == set(), "rows for names no fragment uses (or the mock no longer declares): %s" % sorted(set(table) - used)
    assert len(table) >= 40


def test_the_solvers_objects_are_used_through_the_listed_members_only():
    table = _table()
    code = _code(_read("src", "initCuda.H")) + _code(_read("src", "advect.H"))
    allowed = {"mesh": {"points", "faces", "faceOwner", "faceNeighbour", "nInternalFaces", "nCells"},
               "U": {"primitiveField"}, "runTime": {"value", "deltaT"}, "cudaParticleAdvectionDict": {"getOrDefault"},
               "Pstream": {"nProcs", "myProcNo", "master", "gatherList", "scatterList", "scatter"}}
    for obj, members in allowed.items():
        seen = set(re.findall(r"\b%s(?:\.|::)(\w+)" % obj, code))
        assert seen and seen <= members, (obj, sorted(seen - members))
        assert members <= set(table), (obj, sorted(members - set(table)))
    # (what comes back from those members is used through List's interface and the vector accessors)
    assert set(re.findall(r"\)\.(\w+)\(", code)) <= {"cdata", "size", "value", "x", "y", "z", "min", "max", "c_str", "data"}


def test_the_mock_declares_each_row_the_way_the_table_says():
    """Loose, textual: the distinctive part of the 'mock' column is found in the mock header."""
    mock = re.sub(r"\s+", " ", _read("mock_openfoam", "fvCFD.H"))
    table = _table()
    for name, (decl, _) in table.items():
        assert re.search(r"\b%s\b" % re.escape(name), mock), name
    for needle in ("typedef int label", "typedef double scalar", "typedef vector point", "typedef List<label> labelList",
                   "typedef labelList face", "typedef List<face> faceList", "typedef List<point> pointField",
                   "typedef List<vector> vectorField", "const T* cdata() const", "label size() const",
                   "const pointField& points() const", "const faceList& faces() const", "const labelList& faceOwner() const",
                   "const labelList& faceNeighbour() const", "label nCells() const", "label nInternalFaces() const",
                   "const vectorField& primitiveField() const", "dimensionedScalar deltaT() const",
                   "T getOrDefault(const std::string& k, const T& dflt) const", "static void gatherList(List<T>& l)",
                   "static void scatterList(List<T>& l)", "static void scatter(T& v)", "inline string hostName()",
                   "#define FatalErrorInFunction Foam::FatalStream()"):
        assert needle in mock, needle
