"""GPU: the drop-in boundary end to end.  A mock solver with the reference solver's shape includes the
replacement initCuda.H / advect.H (fused path) and another drives the compat shims in the reference's
five-call order (staged path); both must reproduce the Python host's results bit for bit."""
import glob
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "cudaparticlesfoam_amd", "compat")

DICT = dict(numParticles=20000, dt=1e-4, diffusionCoeff=0.0, saveInterval=10, startTime=0.0, endTime=1e5,
            seedingBox=((-0.02, 0.025, 0.0001), (0.0, 0.0, -0.0001)))       # tutorial box (inverted y/z bounds)
DELTA_T = 35e-4                                                               # 35 Lagrangian cycles


def _expected(pitz, gpu_ctx_factory, D=0.0, cycles=35):
    ctx = gpu_ctx_factory()
    ctx.set_mesh(pitz["mesh"]); ctx.set_velocity(pitz["U_analytic"])
    ctx.seed_box(DICT["numParticles"], *DICT["seedingBox"], 1)
    n_out = ctx.locate_initial()
    ctx.step(DELTA_T / cycles, D, cycles)
    xyzw, cell = ctx.get_particles()
    return xyzw, cell, n_out


def _run(binary, case, cwd):
    subprocess.run(["make", "-C", COMPAT, "-s"], check=True)
    r = subprocess.run([os.path.join(COMPAT, "bin", binary), case], cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    n = DICT["numParticles"]
    xyzw = np.fromfile(os.path.join(cwd, "particles_out.f64")).reshape(n, 4)
    cell = np.fromfile(os.path.join(cwd, "cells_out.i32"), dtype=np.int32)
    return xyzw, cell, r.stdout


def test_replacement_fragments_match_python_host(tmp_path, pitz, gpu_ctx_factory):
    from case_dump import dump_case
    case = str(tmp_path / "case")
    dump_case(case, pitz["mesh"], pitz["U_analytic"], DICT, 1.0, DELTA_T)
    xyzw, cell, out = _run("mockUncoupledFoam", case, str(tmp_path))
    ex, ec, n_out = _expected(pitz, gpu_ctx_factory)
    assert np.array_equal(xyzw, ex) and np.array_equal(cell, ec)
    assert "nCycles: 35" in out and ("Out-of-domain particles(-tetID) = %d" % n_out) in out
    # output cadence of the reference fragment: frame 0, then step+1 for step % saveInterval == 0
    frames = sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / "particle_*.vtu")))
    assert frames == ["particle_%04d.vtu" % k for k in (0, 1, 11, 21, 31)]
    text = open(str(tmp_path / "particle_0031.vtu")).read().splitlines()
    n = DICT["numParticles"]
    assert text[0].startswith("<VTKFile type='UnstructuredGrid'") and text[2] == "<Piece NumberOfCells='%d' NumberOfPoints='%d'>" % (n, n)
    assert len(text) == 10 * n + 32 - 1 + 0 or len(text) > 10 * n     # 10 per-particle arrays
    names = [l.split("Name='")[1].split("'")[0] for l in text if "Name='" in l]
    assert names == ["Position", "ParticleType", "ParticleID", "ParticleTetID", "ConvexTetID", "vels", "KEs",
                     "connectivity", "offsets", "types"]


def test_staged_shims_match_fused_kernel(tmp_path, pitz, gpu_ctx_factory):
    """cudaAdvect -> cudaBrownianMotion -> convexTetQuery -> convexWallReflect -> cudaMoveParticles on the
    reference's AoS arrays == the fused kernel, bit for bit (D = 0 and D > 0: same counter-based stream)."""
    from case_dump import dump_case
    for D in (0.0, 1.5e-5):
        d = dict(DICT, diffusionCoeff=D)
        case = str(tmp_path / ("case%g" % D))
        dump_case(case, pitz["mesh"], pitz["U_analytic"], d, 1.0, DELTA_T)
        wd = tmp_path / ("run%g" % D); wd.mkdir()
        xyzw, cell, out = _run("mockStagedFoam", case, str(wd))
        ex, ec, _ = _expected(pitz, gpu_ctx_factory, D=D)
        assert np.array_equal(cell, ec)
        assert np.array_equal(xyzw[:, :3], ex[:, :3])
        assert os.path.exists(str(wd / "particle_0035.vtu")) and "System Kinetic Energy" in out
