"""Parity against a CONTRACTING build of the reference (round-5 verdict, "Next round" 2).

nvcc fuses a*b+c into one fma by default (--fmad=true) and the reference's build does not switch that off; the strict build
of the reference's functions that pins the oracle (oracle/_ref/libref_rtxadvect.so, g++ -ffp-contract=off) therefore rounds
differently from the binary the authors ran.  oracle/build_ref.sh builds the same splices a second time with
-ffp-contract=fast -mfma (libref_rtxadvect_fma.so), tests/golden/make_golden.py fma stores what THAT build produces
(tests/golden/*_fma.npz) and what separates the two builds on 400 000 particles (profiles/r06_fma_sensitivity.json).

CPU: the committed measurement is re-taken where both builds exist (this container and, the .so files travelling, the GPU box),
the fma goldens are reproduced by the contracting build bit for bit, and the C restatement of the cell walk -- what the HIP
kernels implement -- meets the contract against them: cells equal, |dx|/L <= 1e-5 for >= 99.99 % of the particles, outliers
counted and printed.  GPU: the HIP path against the same fma goldens, and live against the
contracting build on 100 000 particles over 1000 cycles."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = os.path.join(HERE, "golden")
sys.path.insert(0, G)

CASES = ["pitz_uniform", "pitz_analytic", "box_random"]
REL_TOL, QUANTILE = 1e-5, 0.9999


def _mg():
    import make_golden as mg
    return mg


def _fma_inputs(name, oracle_libs):
    mg = _mg()
    g = np.load(os.path.join(G, name + "_fma.npz"))
    n = g["xyz0"].shape[0]
    mesh, centres, U, xyz, cell0, L = mg.fma_case_inputs(name, n, oracle_libs.CellWalk())
    assert str(g["inputs_sha256"]) == mg.digest(mesh.points, mesh.face_verts, mesh.owner, mesh.neighbour, U), \
        "synthetic inputs drifted from the ones the fma goldens were generated with"
    assert np.array_equal(xyz, g["xyz0"]) and np.array_equal(cell0, g["tet0"] // 12)
    return g, mesh, centres, U, L


def _contract(rel, cells_differ, what):
    """The bar of BASELINE.md section 3, with the outliers counted and printed."""
    out = int((rel > REL_TOL).sum())
    print("%s: %d particles, %d beyond %g (%.4f %%), %d cells differ, max |dx|/L %.3e, 99.99-percentile %.3e"
          % (what, rel.size, out, REL_TOL, 100.0 * out / rel.size, cells_differ, rel.max(), np.quantile(rel, QUANTILE)))
    assert out <= (1.0 - QUANTILE) * rel.size, what
    assert cells_differ <= (1.0 - QUANTILE) * rel.size, what


def test_committed_sensitivity_report_says_what_the_design_quotes():
    r = json.load(open(os.path.join(ROOT, "profiles", "r06_fma_sensitivity.json")))
    assert set(r["cases"]) == set(CASES)
    for name, rows in r["cases"].items():
        assert [row["cycles"] for row in rows] == ([1, 10, 100, 1000] if name.startswith("pitz") else [1, 20, 100])
        for row in rows:
            assert row["particles"] == 400000
            assert row["cells_differ"] == 0 and row["tets_differ"] == 0 and row["beyond_1e5"] == 0
            assert row["max_rel"] < 1e-13                       # contraction moves a particle by rounding errors, not cells


@pytest.mark.parametrize("name", CASES)
def test_strict_against_contracting_build_of_the_reference(name, oracle_libs):
    """The measurement itself, re-taken here on 20 000 particles: fraction of particles in a different cell and max /
    99.99-percentile |dx|/L between the two builds of the reference's own functions at every checkpoint."""
    if not (oracle_libs.have_ref() and oracle_libs.have_ref_fma()):
        pytest.skip("oracle/_ref (strict + contracting) not built here (needs /root/reference at build time)")
    mg = _mg()
    n, dt, cps = 20000, *[(c[2], c[3]) for c in mg.FMA_CASES if c[0] == name][0]
    mesh, centres, U, xyz, cell0, L = mg.fma_case_inputs(name, n, oracle_libs.CellWalk())
    a = mg.run_case(oracle_libs.RefLib(), mesh, centres, U, xyz, cell0, dt, cps)
    b = mg.run_case(oracle_libs.RefLib(fma=True), mesh, centres, U, xyz, cell0, dt, cps)
    for row in mg.compare_builds(a, b, L, cps):
        print(name, row)
        assert row["cells_differ"] <= 1e-4 * n and row["p9999_rel"] <= REL_TOL and row["beyond_1e5"] <= 1e-4 * n
        assert row["max_rel"] < 1e-12                           # measured: <= 4e-15 (profiles/r06_fma_sensitivity.json)


@pytest.mark.parametrize("name", CASES)
def test_contracting_build_reproduces_its_goldens_bitwise(name, oracle_libs):
    if not oracle_libs.have_ref_fma():
        pytest.skip("oracle/_ref/libref_rtxadvect_fma.so not built here")
    g, mesh, centres, U, L = _fma_inputs(name, oracle_libs)
    mg = _mg()
    out = mg.run_case(oracle_libs.RefLib(fma=True), mesh, centres, U, g["xyz0"], (g["tet0"] // 12).astype(np.int32),
                      float(g["dt"]), [int(k) for k in g["checkpoints"]])
    for k in g["checkpoints"]:
        assert np.array_equal(out["P_%d" % k][:, :3], g["P_%d" % k]) and np.array_equal(out["tet_%d" % k], g["tet_%d" % k])


@pytest.mark.parametrize("name", CASES)
def test_cellwalk_meets_the_contract_against_the_fma_goldens(name, oracle_libs):
    """oracle/cellwalk.c (bit-identical to the HIP kernels, tests/test_gpu_parity.py) against what the contracting build of
    the reference produced."""
    g, mesh, centres, U, L = _fma_inputs(name, oracle_libs)
    cw = oracle_libs.CellWalk()
    t = cw.build(mesh)
    x, y, z = (g["xyz0"][:, k].copy() for k in range(3))
    c = (g["tet0"] // 12).astype(np.int32)
    done = 0
    for k in g["checkpoints"]:
        cw.step(x, y, z, c, float(g["dt"]), int(k) - done, t, U, nthreads=cw.max_threads)
        done = int(k)
        P = g["P_%d" % k]
        rel = np.sqrt((x - P[:, 0]) ** 2 + (y - P[:, 1]) ** 2 + (z - P[:, 2]) ** 2) / L
        _contract(rel, int((c != g["tet_%d" % k] // 12).sum()), "%s k=%d cellwalk.c vs fma golden" % (name, k))
        assert rel.max() < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_meets_the_contract_against_the_fma_goldens(name, oracle_libs, gpu_ctx_factory):
    g, mesh, centres, U, L = _fma_inputs(name, oracle_libs)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(U)
    ctx.set_particles(g["xyz0"], (g["tet0"] // 12).astype(np.int32))
    done = 0
    for k in g["checkpoints"]:
        ctx.step(float(g["dt"]), 0.0, int(k) - done)
        done = int(k)
        xyzw, cell = ctx.get_particles()
        P = g["P_%d" % k]
        rel = np.sqrt(((xyzw[:, :3] - P) ** 2).sum(1)) / L
        _contract(rel, int((cell != g["tet_%d" % k] // 12).sum()), "%s k=%d HIP vs fma golden" % (name, k))
        assert rel.max() < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pitz_analytic", "box_random"])
def test_hip_against_the_contracting_reference_live_on_1e5_particles(name, oracle_libs, gpu_ctx_factory):
    """The same bar on a sample large enough for the 99.99 % to mean something: 100 000 particles, the reference's own
    functions (contracting build, on the host cores) against the HIP path at every checkpoint up to 1000 / 100 cycles."""
    if not oracle_libs.have_ref_fma():
        pytest.skip("oracle/_ref/libref_rtxadvect_fma.so did not travel to this box")
    mg = _mg()
    n, dt, cps = 100000, *[(c[2], c[3]) for c in mg.FMA_CASES if c[0] == name][0]
    mesh, centres, U, xyz, cell0, L = mg.fma_case_inputs(name, n, oracle_libs.CellWalk())
    ref = mg.run_case(oracle_libs.RefLib(fma=True), mesh, centres, U, xyz, cell0, dt, cps)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(U)
    ctx.set_particles(xyz, cell0.astype(np.int32))
    done = 0
    for k in cps:
        ctx.step(dt, 0.0, k - done)
        done = k
        xyzw, cell = ctx.get_particles()
        rel = np.sqrt(((xyzw[:, :3] - ref["P_%d" % k][:, :3]) ** 2).sum(1)) / L
        _contract(rel, int((cell != ref["tet_%d" % k] // 12).sum()), "%s k=%d HIP vs contracting reference, live" % (name, k))
