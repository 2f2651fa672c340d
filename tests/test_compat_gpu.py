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


def _expected(pitz, gpu_ctx_factory, D=0.0, cycles=35, z_fold=1):
    ctx = gpu_ctx_factory()
    ctx.set_option("z_fold", z_fold)
    ctx.set_mesh(pitz["mesh"]); ctx.set_velocity(pitz["U_analytic"])
    ctx.seed_box(DICT["numParticles"], *DICT["seedingBox"], 1)
    n_out = ctx.locate_initial()
    ctx.step(DELTA_T / cycles, D, cycles)
    xyzw, cell = ctx.get_particles()
    return xyzw, cell, n_out


def _run(binary, case, cwd, n=None, extra=(), env=None):
    subprocess.run(["make", "-C", COMPAT, "-s"], check=True)
    r = subprocess.run([os.path.join(COMPAT, "bin", binary), case] + list(extra), cwd=cwd, capture_output=True, text=True,
                       timeout=600, env=None if env is None else dict(os.environ, **env))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    n = DICT["numParticles"] if n is None else n
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


def _vtu_array(path, name, width):
    lines = open(path).read().splitlines()
    k = next(i for i, l in enumerate(lines) if "Name='%s'" % name in l)
    rows = []
    for l in lines[k + 1:]:
        if l.startswith("<"):
            break
        rows.append([float(v) for v in l.split()])
    return np.array(rows).reshape(-1, width)


def test_frame_zero_carries_velocities_and_inactive_particles(tmp_path, pitz):
    """particle_0000.vtu like the reference's (src/initCuda.H:184-201 runs one cudaAdvect before the first frame):
    vels = U[cell], and a particle seeded outside the mesh is ParticleType 0 (cuda/particles.cu:333-338)."""
    from case_dump import dump_case
    d = dict(DICT, numParticles=3000, seedingBox=((-0.0215, 0.03, 0.0001), (0.0, 0.0, -0.0001)))   # sticks out of the inlet
    case = str(tmp_path / "case")
    dump_case(case, pitz["mesh"], pitz["U_analytic"], d, 1.0, 1e-4)
    xyzw, cell, out = _run("mockUncoupledFoam", case, str(tmp_path), n=3000)
    f0 = str(tmp_path / "particle_0000.vtu")
    vels = _vtu_array(f0, "vels", 3); ptype = _vtu_array(f0, "ParticleType", 1)[:, 0]
    pos = _vtu_array(f0, "Position", 3)
    n_out = int(out.split("Out-of-domain particles(-tetID) = ")[1].split()[0])
    assert 0 < n_out < 3000 and int((ptype == 0).sum()) == n_out
    inside = ptype == 1
    assert np.abs(vels[inside]).max() > 1.0 and np.abs(vels[~inside]).max() == 0.0
    # U[cell] of the frozen step flow at the particles' own positions: compare with the host-side field lookup
    from cudaparticlesfoam_amd.api import Context
    with Context(0) as ctx:
        ctx.set_mesh(pitz["mesh"]); ctx.set_velocity(pitz["U_analytic"])
        ctx.set_particles(pos); ctx.locate_initial()
        _, c0 = ctx.get_particles()
    assert np.array_equal(c0 >= 0, inside)
    assert np.allclose(vels[inside], pitz["U_analytic"][c0[inside]], rtol=0, atol=5e-7)      # "%.6f"-style precision


@pytest.mark.parametrize("n_procs", [2, 4])
def test_parallel_fragments_equal_the_serial_run(tmp_path, pitz, n_procs):
    """The Pstream branch of the fragments (replacing src/initCuda.H:207-484, src/advect.H:59-84) is RANK PER GPU: every
    rank gets every piece of the decomposed mesh (gatherList + scatterList), stitches them (cpf_set_mesh_parts), creates its
    own context and shard, joins one communicator (token over Pstream::scatter) and owns a cell range; particles change hands
    between the ranks' shards, frames are gathered to the master.  One process plays the ranks as threads on the one GPU (mock
    Pstream, in-process communicator); particles and output frames must equal the serial run's byte for byte."""
    from case_dump import dump_case
    from cudaparticlesfoam_amd.cases import split_into_parts
    mesh, U = pitz["mesh"], pitz["U_analytic"]
    case = str(tmp_path / "case")
    dump_case(case, mesh, U, DICT, 1.0, DELTA_T)
    ser = tmp_path / "serial"; ser.mkdir()
    xs, cs, outs = _run("mockUncoupledFoam", case, str(ser))
    parts = split_into_parts(mesh, n_procs)
    first = 0
    for r, part in enumerate(parts):
        dump_case(os.path.join(case, "processor%d" % r), part, U[first:first + part.n_cells], DICT, 1.0, DELTA_T)
        first += part.n_cells
    par = tmp_path / "parallel"; par.mkdir()
    xp, cp, outp = _run("mockParallelFoam", case, str(par), extra=[str(n_procs)])
    assert np.array_equal(xp, xs) and np.array_equal(cp, cs)
    assert "nCycles: 35" in outp and ("%d GPUs" % n_procs) in outp
    assert outs.split("Out-of-domain particles(-tetID) = ")[1].split()[0] == outp.split("Out-of-domain particles(-tetID) = ")[1].split()[0]
    handed = int(outp.split(" re-cuts, ")[1].split()[0])
    assert handed > 0                                                     # particles did change hands between the ranks
    frames = sorted(os.path.basename(p) for p in glob.glob(str(par / "particle_*.vtu")))
    assert frames == ["particle_%04d.vtu" % k for k in (0, 1, 11, 21, 31)]
    for f in frames:
        assert open(str(par / f), "rb").read() == open(str(ser / f), "rb").read(), f


@pytest.mark.parametrize("how", ["more ranks than GPUs", "rankPerGPU false"])
def test_more_ranks_than_gpus_fall_back_to_the_reference_topology(tmp_path, pitz, how):
    """Round-5 advisory (high): `mpirun -np N` with N > the GPUs of a host must not abort in ncclCommInitRank ("Duplicate GPU").
    The fragments count ranks and GPUs per host (hostName + cpf_device_count over gatherList) and, when a host has fewer GPUs
    than ranks -- here 3 rank-threads with the RCCL kind on this box's GPUs, or the dictionary key `rankPerGPU false` -- fall
    back to the reference's own parallel topology (src/initCuda.H:207-484, src/advect.H:59-89): the master stitches the pieces
    and drives ONE GPU with the whole cloud, the other ranks contribute their mesh piece and, every Eulerian step, their U
    slice through gatherList.  Two Eulerian steps of a transient field with the tutorial's diffusion: particles and frames
    equal the serial run's byte for byte."""
    import torch
    from case_dump import dump_case
    from cudaparticlesfoam_amd.cases import split_into_parts
    n_procs = torch.cuda.device_count() + 2
    mesh, U = pitz["mesh"], pitz["U_analytic"]
    d = dict(DICT, diffusionCoeff=1.5e-5)
    env = {"CPF_COMM": "rccl"}
    if how == "rankPerGPU false":
        d["rankPerGPU"] = 0
        env = {"CPF_COMM": "inprocess"}             # (would allow rank-threads to share the device: the key must win)
    case = str(tmp_path / "case")
    dump_case(case, mesh, U, d, 1.0, DELTA_T)
    ser = tmp_path / "serial"; ser.mkdir()
    xs, cs, outs = _run("mockUncoupledFoam", case, str(ser), extra=["2"])
    first = 0
    for r, part in enumerate(split_into_parts(mesh, n_procs)):
        dump_case(os.path.join(case, "processor%d" % r), part, U[first:first + part.n_cells], d, 1.0, DELTA_T)
        first += part.n_cells
    par = tmp_path / "parallel"; par.mkdir()
    xp, cp, outp = _run("mockParallelFoam", case, str(par), extra=[str(n_procs), "2"], env=env)
    assert "the master drives one GPU" in outp and "drives the one GPU" in outp and "GPUs)" not in outp
    assert np.array_equal(xp, xs) and np.array_equal(cp, cs)
    frames = sorted(os.path.basename(p) for p in glob.glob(str(par / "particle_*.vtu")))
    assert frames == sorted(os.path.basename(p) for p in glob.glob(str(ser / "particle_*.vtu"))) and len(frames) >= 8
    for f in frames:
        assert open(str(par / f), "rb").read() == open(str(ser / f), "rb").read(), f


@pytest.mark.parametrize("n_procs", [2, 4])
def test_parallel_fragments_transient_field_with_diffusion(tmp_path, pitz, n_procs):
    """Three Eulerian steps of a transient solver with the tutorial's diffusion coefficient: every rank uploads its own U
    slice each step (cpf_shard_set_velocity_slice: all-gathered between the shards), the counter-based Brownian stream is
    keyed by (seed, particle id, cycle) so a particle draws the same kicks on whichever rank holds it, and frame 0 does not
    count as a cycle -- particles and frames equal the serial run's byte for byte."""
    from case_dump import dump_case
    from cudaparticlesfoam_amd.cases import split_into_parts
    mesh, U = pitz["mesh"], pitz["U_analytic"]
    d = dict(DICT, diffusionCoeff=1.5e-5, saveInterval=7, saveStreamline=1)        # + the trajectory files (src/advect.H:163-175)
    case = str(tmp_path / "case")
    dump_case(case, mesh, U, d, 1.0, 20e-4)
    ser = tmp_path / "serial"; ser.mkdir()
    xs, cs, outs = _run("mockUncoupledFoam", case, str(ser), extra=["3"])
    first = 0
    for r, part in enumerate(split_into_parts(mesh, n_procs)):
        dump_case(os.path.join(case, "processor%d" % r), part, U[first:first + part.n_cells], d, 1.0, 20e-4)
        first += part.n_cells
    par = tmp_path / "parallel"; par.mkdir()
    xp, cp, outp = _run("mockParallelFoam", case, str(par), extra=[str(n_procs), "3"])
    assert np.array_equal(xp, xs) and np.array_equal(cp, cs)
    assert outp.count("nCycles: 20") == 3
    frames = sorted(os.path.basename(p) for p in glob.glob(str(par / "particle_*.vtu")))
    assert frames == sorted(os.path.basename(p) for p in glob.glob(str(ser / "particle_*.vtu"))) and len(frames) >= 9
    for f in frames + ["Streamline.vtk"]:
        assert open(str(par / f), "rb").read() == open(str(ser / f), "rb").read(), f
    head = open(str(ser / "Streamline.vtk")).read(400)
    assert head.startswith("# vtk DataFile Version 4.1") and "POINTS" in head


@pytest.mark.parametrize("seed", range(int(os.environ.get("CPF_FUZZ_DICTS", "6"))))        # (a longer campaign: CPF_FUZZ_DICTS=80)
def test_random_dictionaries_parallel_equals_serial(tmp_path, pitz, seed):
    """Random cudaParticlesDict entries (particle count, diffusion on / off, output cadence, a seeding box that may stick out of
    the domain, ASCII or binary frames, trajectories), cycles per Eulerian step, Eulerian steps and rank counts: the rank-per-GPU
    run of the replacement fragments (ranks as threads on the one GPU) equals the serial run -- final particles, every frame and
    the trajectory file byte for byte.  The cycles between two frames go through cpf_shard_step with CPF_STEP_FUSE_CYCLES, with the
    fragments' own cadences (re-cut every 32 cycles by measured time, derived overlap depth, sorts every 25 / 50)."""
    from case_dump import dump_case
    from cudaparticlesfoam_amd.cases import split_into_parts
    rng = np.random.default_rng(4200 + seed)
    mesh, U = pitz["mesh"], pitz["U_analytic"]
    n_procs = int(rng.integers(2, 6))
    cycles, esteps = int(rng.integers(5, 70)), int(rng.integers(1, 4))
    box = DICT["seedingBox"] if rng.integers(0, 2) else ((-0.025, -0.03, -0.0003), (0.05, 0.03, 0.0003))    # partly outside
    d = dict(DICT, numParticles=int(rng.integers(3000, 40000)), diffusionCoeff=float(rng.choice([0.0, 1.5e-5])),
             saveInterval=int(rng.integers(1, 15)), seedingBox=box, binaryFrames=int(rng.integers(0, 2)),
             saveStreamline=int(rng.integers(0, 3) == 0))
    case = str(tmp_path / "case")
    dump_case(case, mesh, U, d, 1.0, cycles * 1e-4)
    ser = tmp_path / "serial"; ser.mkdir()
    xs, cs, outs = _run("mockUncoupledFoam", case, str(ser), n=d["numParticles"], extra=[str(esteps)])
    first = 0
    for r, part in enumerate(split_into_parts(mesh, n_procs)):
        dump_case(os.path.join(case, "processor%d" % r), part, U[first:first + part.n_cells], d, 1.0, cycles * 1e-4)
        first += part.n_cells
    par = tmp_path / "parallel"; par.mkdir()
    xp, cp, outp = _run("mockParallelFoam", case, str(par), n=d["numParticles"], extra=[str(n_procs), str(esteps)])
    assert np.array_equal(xp, xs) and np.array_equal(cp, cs), (seed, d, n_procs, cycles, esteps)
    assert outp.count("nCycles: ") == esteps and outs.count("nCycles: ") == esteps       # (deltaT / dt may round up by one cycle)
    frames = sorted(os.path.basename(p) for p in glob.glob(str(par / "particle_*.vtu")))
    assert frames == sorted(os.path.basename(p) for p in glob.glob(str(ser / "particle_*.vtu"))) and len(frames) >= 2
    for f in frames + (["Streamline.vtk"] if d["saveStreamline"] else []):
        assert open(str(par / f), "rb").read() == open(str(ser / f), "rb").read(), (f, seed, d, n_procs, cycles, esteps)


def test_parallel_fragments_on_an_irregular_decomposition(tmp_path, pitz):
    """What scotch hands the ranks: every cell goes to the nearest of four random centres (pieces of unequal size with ragged
    cuts).  The stitched mesh numbers the cells piece by piece, so cell ids come out permuted and a cut face may be oriented the
    other way round (its plane then comes from the reversed vertex loop): the same particles as the serial run, cells equal
    under the renumbering, positions equal to rounding."""
    from case_dump import dump_case
    from cudaparticlesfoam_amd.cases import split_into_parts
    mesh, U = pitz["mesh"], pitz["U_analytic"]
    rng = np.random.default_rng(99)
    n_procs = 4
    centres, _ = mesh.cell_centres_volumes()
    pick = centres[rng.choice(mesh.n_cells, size=n_procs, replace=False)]
    cell_part = np.argmin(((centres[:, None, :2] - pick[None, :, :2]) ** 2).sum(axis=2), axis=1)
    order = np.lexsort((np.arange(mesh.n_cells), cell_part))                       # order[new] = old
    new_of_old = np.empty(mesh.n_cells, np.int64); new_of_old[order] = np.arange(mesh.n_cells)
    case = str(tmp_path / "case")
    dump_case(case, mesh, U, DICT, 1.0, DELTA_T)
    ser = tmp_path / "serial"; ser.mkdir()
    xs, cs, outs = _run("mockUncoupledFoam", case, str(ser))
    for r, part in enumerate(split_into_parts(mesh, n_procs, cell_part)):
        dump_case(os.path.join(case, "processor%d" % r), part, U[cell_part == r], DICT, 1.0, DELTA_T)
    par = tmp_path / "parallel"; par.mkdir()
    xp, cp, outp = _run("mockParallelFoam", case, str(par), extra=[str(n_procs)])
    assert ("%d GPUs" % n_procs) in outp
    live = cs >= 0
    assert np.array_equal(live, cp >= 0)
    same = new_of_old[np.maximum(cs, 0)] == cp
    assert same[live].mean() > 0.999
    assert np.abs(xp - xs)[live & same].max() < 1e-12
    assert int(outp.split(" re-cuts, ")[1].split()[0]) > 0                 # particles did change hands


def test_staged_shims_match_fused_kernel(tmp_path, pitz, gpu_ctx_factory):
    """cudaAdvect -> cudaBrownianMotion -> convexTetQuery -> convexWallReflect -> cudaMoveParticles on the
    reference's AoS arrays == the fused kernel, bit for bit (D = 0 and D > 0: same counter-based stream).  The stages
    keep the reference's order of operations; with the kick on this one-cell-thick mesh the fused kernel by default
    mirrors the end point about the front / back plane BEFORE the walk (cpf_walk.h, fold_z): bit-identical with that
    switched off, and the same trajectory to rounding with it on."""
    from case_dump import dump_case
    for D in (0.0, 1.5e-5):
        d = dict(DICT, diffusionCoeff=D)
        case = str(tmp_path / ("case%g" % D))
        dump_case(case, pitz["mesh"], pitz["U_analytic"], d, 1.0, DELTA_T)
        wd = tmp_path / ("run%g" % D); wd.mkdir()
        xyzw, cell, out = _run("mockStagedFoam", case, str(wd))
        ex, ec, _ = _expected(pitz, gpu_ctx_factory, D=D, z_fold=0)
        assert np.array_equal(cell, ec)
        assert np.array_equal(xyzw[:, :3], ex[:, :3])
        fx, fc, _ = _expected(pitz, gpu_ctx_factory, D=D)                      # the default: end point mirrored before the walk
        assert (fc == ec).mean() > 0.9995 and np.abs(fx[:, :3] - ex[:, :3])[fc == ec].max() < 1e-12
        assert os.path.exists(str(wd / "particle_0035.vtu")) and "System Kinetic Energy" in out
        assert "#mock: 20.00K particles, 35 cycles (convexTetQuery + convexWallReflect)" in out       # (prettyNumber, cudaTimer)
        # the reference's RTX branch of the cycle (src/advect.H:126-135): RTQuery in displacement mode + RTWallReflect -- whose disps /
        # vels arguments come in the other order -- resolve to the same plane walk: the same bits
        wr = tmp_path / ("rtx%g" % D); wr.mkdir()
        rx, rc, rout = _run("mockStagedFoam", case, str(wr), extra=["rtx"])
        assert np.array_equal(rx, xyzw) and np.array_equal(rc, cell) and "(RTQuery + RTWallReflect)" in rout
        assert open(str(wr / "particle_0035.vtu"), "rb").read() == open(str(wd / "particle_0035.vtu"), "rb").read()


def test_tjunction_allrun_parallel_equals_serial(tmp_path):
    """The reference's own parallel tutorial run (TJunction/Allrun-parallel:9-12: decomposePar simple (4 1 1), then
    `mpirun -np 4 cudaParticlesPimpleFoam -parallel`) through the replacement fragments' rank-per-GPU branch: the mesh cut
    into four x slabs of equal cell count, four ranks (threads here) each stitching the pieces, owning a cell range of the
    cloud and handing particles over -- with the tutorial's diffusion.  Particles and frames equal the serial run's byte for byte."""
    from case_dump import dump_case
    from cudaparticlesfoam_amd.cases import split_into_parts
    from cudaparticlesfoam_amd.cases import tjunction as tj
    from cudaparticlesfoam_amd.parallel import x_slab_renumbering
    m0 = tj.tjunction_mesh(); c0, _ = m0.cell_centres_volumes()
    mesh = m0.renumber_cells(x_slab_renumbering(c0))              # contiguous cell ranges == x slabs, like `simple (4 1 1)`
    centres, _ = mesh.cell_centres_volumes()
    U = tj.split_flow_u(mesh, centres, 0.5, u0=5.0)
    d = dict(tj.PARTICLE_DICT, numParticles=50000, endTime=10.0)
    case = str(tmp_path / "case")
    dump_case(case, mesh, U, d, 0.5, tj.EULERIAN_DT)
    ser = tmp_path / "serial"; ser.mkdir()
    xs, cs, outs = _run("mockUncoupledFoam", case, str(ser), n=d["numParticles"])
    parts = split_into_parts(mesh, 4)
    first = 0
    for r, part in enumerate(parts):
        dump_case(os.path.join(case, "processor%d" % r), part, U[first:first + part.n_cells], d, 0.5, tj.EULERIAN_DT)
        first += part.n_cells
    par = tmp_path / "parallel"; par.mkdir()
    xp, cp, outp = _run("mockParallelFoam", case, str(par), n=d["numParticles"], extra=["4"])
    assert np.array_equal(xp, xs) and np.array_equal(cp, cs)
    assert "nCycles: 10" in outp and (cs >= 0).all()
    frames = sorted(os.path.basename(p) for p in glob.glob(str(par / "particle_*.vtu")))
    assert frames == ["particle_%04d.vtu" % k for k in (0, 1, 3, 5, 7, 9)]
    for f in frames:
        assert open(str(par / f), "rb").read() == open(str(ser / f), "rb").read(), f
