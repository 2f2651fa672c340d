"""CPU: the replacement fragments + compat shims compile and link against the mock OpenFOAM types
with plain g++ (no HIP headers on the solver side), and the resulting solver fails loudly without a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "cudaparticlesfoam_amd", "compat")


def test_fragments_and_shims_compile_and_link():
    r = subprocess.run(["make", "-C", COMPAT, "-B", "-s"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    for b in ("mockUncoupledFoam", "mockStagedFoam", "mockParallelFoam"):
        assert os.access(os.path.join(COMPAT, "bin", b), os.X_OK)
    # the parallel (Pstream) branch of the fragments is compiled code: it binds the rank-direct ingest
    par = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(COMPAT, "bin", "mockParallelFoam")],
                         capture_output=True, text=True, check=True).stdout
    assert "cpf_set_mesh_parts" in par and "cpf_set_velocity" in par
    out = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(COMPAT, "bin", "mockUncoupledFoam")],
                         capture_output=True, text=True, check=True).stdout
    undefined = {l.split()[-1] for l in out.splitlines()}
    assert {"cpf_create", "cpf_set_mesh", "cpf_set_velocity", "cpf_seed_box", "cpf_locate_initial", "cpf_step",
            "cpf_write_vtu"} <= undefined
    assert not any(s.startswith("hip") for s in undefined)       # the solver side never touches HIP itself


def test_mock_solver_fails_loudly_without_gpu(tmp_path, pitz):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from case_dump import dump_case
    dump_case(str(tmp_path / "case"), pitz["mesh"], pitz["U_uniform"], dict(numParticles=10), 0.0, 1e-4)
    subprocess.run(["make", "-C", COMPAT, "-s"], check=True)
    r = subprocess.run([os.path.join(COMPAT, "bin", "mockUncoupledFoam"), str(tmp_path / "case")], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr
