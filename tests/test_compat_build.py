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


def test_reference_io_block_compiles_against_the_shims(tmp_path):
    """A host that kept the reference's per-cycle I/O block verbatim (src/advect.H:163-175: addToTrajectories,
    writeParticles2VTU, saveTrajectories, writeStreamline2VTK behind saveStreamlinetoFile) and all three cudaAdvect mode
    strings (cuda/particles.cu:417-445) compiles against compat/cuda/common.h with plain g++; the two trajectory writers --
    dead in the reference, src/initCuda.H:68 -- run on the host's own vectors (cpf_traj_*_arrays): trajectories with fewer than
    two points are left out, OBJ vertex numbers are 1-based per file."""
    src = tmp_path / "io_block.cpp"
    src.write_text(r'''
#include "cuda/common.h"
#include <cstring>
using namespace advect;
int main() {
    Particle* d_particles = nullptr; vec4d* d_particle_vels = nullptr; vec4d* d_disp = nullptr;
    int* d_particles_tetIDs = nullptr; int* d_particles_ConvextetIDs = nullptr;
    int numParticles = 0, step = 0, saveInterval = 10;
    bool saveStreamlinetoFile = false;
    std::string objTrajectoryFileName, vtkStreamlineFileName;
    std::vector<std::vector<vec3f>> trajectories;
    if (saveStreamlinetoFile)
        if ((step % (saveInterval * 1)) == 0)
            addToTrajectories(d_particles, numParticles, trajectories);
    if (saveStreamlinetoFile) {
        if (objTrajectoryFileName.size() > 0) saveTrajectories(objTrajectoryFileName, trajectories);
        if (vtkStreamlineFileName.size() > 0) writeStreamline2VTK(vtkStreamlineFileName, trajectories);
    }
    void (*advectFn)(Particle*, int*, vec4d*, vec4d*, double, int, vec4i*, vec3d*, vec3d*, std::string) = &cudaAdvect;
    (void)advectFn; (void)d_particle_vels; (void)d_disp; (void)d_particles_tetIDs; (void)d_particles_ConvextetIDs;
    trajectories.resize(3);
    trajectories[0] = {vec3f{0.f, 0.f, 0.f}, vec3f{1.f, 0.5f, 0.25f}, vec3f{2.f, 1.f, 0.5f}};
    trajectories[1] = {vec3f{9.f, 9.f, 9.f}};                    // one sample: in neither file
    trajectories[2] = {vec3f{-1.f, 0.f, 1e-7f}, vec3f{-2.f, 0.f, 123456.789f}};
    saveTrajectories("t.obj", trajectories);
    writeStreamline2VTK("t.vtk", trajectories);
    return 0;
}
''')
    exe = tmp_path / "io_block"
    lib = os.path.join(ROOT, "cudaparticlesfoam_amd", "lib")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + COMPAT, "-I" + os.path.join(ROOT, "include"), str(src),
                        "-L" + lib, "-lcudaParticleAdvection", "-Wl,-rpath," + lib, "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([str(exe)], cwd=str(tmp_path)).returncode == 0
    obj = open(tmp_path / "t.obj").read().splitlines()
    assert obj == ["v 0 0 0", "v 1 0.5 0.25", "v 2 1 0.5", "l 1 2", "l 2 3", "v -1 0 1e-07", "v -2 0 123457", "l 4 5"]
    vtk = open(tmp_path / "t.vtk").read()
    assert "POINTS 5 float" in vtk and "LINES 2 7" in vtk and "\n3 0 1 2\n2 3 4\n" in vtk and "StreamlineID 1 2 int" in vtk
