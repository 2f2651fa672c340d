"""GPU: the HIP path, through the C-ABI, against the COMMITTED reference goldens (tests/golden/*.npz, produced by
the reference's own functions -- tests/golden/make_golden.py), with nothing in between: no freshly built oracle
is trusted here before it, too, has reproduced the same goldens in this very run (last section).

Bars (BASELINE.json): cell ids identical to ``tet // 12`` of the reference for EVERY particle; positions
|dx| / L <= 1e-5 with L = domain bbox diagonal for EVERY particle; seeding bit-exact.
Reference being matched: query/ConvexQuery.cu:135-216 (locate, wall code ``-(start+1)`` at :204-215), :320-436
(reflect), cuda/particles.cu:316-373 (advect), :659-704 (move), :78-97 (seeding).
"""
import os

import numpy as np
import pytest

import test_oracle_golden as og

pytestmark = pytest.mark.gpu
REL_TOL = 1e-5
G = og.G


def _rel(a, b, L):
    return np.sqrt(((a[:, :3] - b[:, :3]) ** 2).sum(axis=1)) / L


@pytest.mark.parametrize("variant", [4, 3, 0])
@pytest.mark.parametrize("name", ["pitz_uniform", "pitz_analytic", "box_random"])
def test_cycles_vs_reference_goldens(name, variant, pitz, gpu_ctx_factory):
    """Inject the golden's start state, run to every checkpoint, compare with what the reference's own
    particleAdvectKernelTetVel / particleLocator / convexReflector / particleMoveKernel produced."""
    from cudaparticlesfoam_amd import _lib as L
    g, mesh, centres, U = og._case_inputs(name, pitz)
    lo, hi = mesh.bounds(); diag = float(np.linalg.norm(hi - lo))
    ctx = gpu_ctx_factory()
    ctx.set_option("step_variant", variant)
    ctx.set_mesh(mesh); ctx.set_velocity(U)
    ctx.set_particles(g["xyz0"], (g["tet0"] // 12).astype(np.int32))
    done, worst = 0, 0.0
    for k in g["checkpoints"]:
        k = int(k)
        if k - done > 1:
            ctx.step(float(g["dt"]), 0.0, k - done - 1)
        ctx.step(float(g["dt"]), 0.0, 1, L.STEP_STORE_VEL)          # the last cycle also stores the velocity
        done = k
        xyzw, cell, vel = ctx.get_particles(want_vel=True)
        P, tet, rv = g["P_%d" % k], g["tet_%d" % k], g["vel_%d" % k]
        assert np.array_equal(cell, tet // 12), "k=%d: %d cells differ" % (k, int((cell != tet // 12).sum()))
        rel = _rel(xyzw, P, diag)
        worst = max(worst, float(rel.max()))
        assert rel.max() <= REL_TOL, "k=%d max |dx|/L = %.3e" % (k, rel.max())
        assert np.array_equal(xyzw[:, 3], P[:, 3])                    # active flag w
        # velocity after the cycle (mirrored by reflections): same cell-constant U, so equal to rounding
        scale = max(1.0, float(np.abs(rv[:, :3]).max()))
        assert np.abs(vel[:, :3] - rv[:, :3]).max() <= 1e-9 * scale
        assert np.array_equal(vel[:, 3], rv[:, 3])                    # w = -1 (particles.cu:361)
    assert worst < 1e-10                                              # in fact orders of magnitude inside the bar


def test_stage_by_stage_vs_reference_goldens(gpu_ctx_factory):
    """cpf_stage_advect / locate / reflect / move on the reference's AoS layouts against every intermediate array
    of one reference cycle, including the wall code -(startCell+1) between locate and reflect."""
    from cudaparticlesfoam_amd.api import StagedCloud
    from cudaparticlesfoam_amd.cases import box_mesh
    g = np.load(os.path.join(G, "stages_box.npz"))
    mesh = box_mesh(10, 9, 8)
    lo, hi = mesh.bounds(); diag = float(np.linalg.norm(hi - lo))
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(g["U"])
    n = g["xyz0"].shape[0]
    cell0 = (g["tet0"] // 12).astype(np.int32)
    P0 = np.ones((n, 4)); P0[:, :3] = g["xyz0"]
    sc = StagedCloud(ctx, n)
    try:
        sc.set(P0, cell0)
        # ---- cudaAdvect: disp = (P + dt U) - P, vels.w = disps.w = -1.  The kernels evaluate P + dt*U as ONE fma
        # (what nvcc's default -fmad=true makes of cuda/particles.cu:357-359); the goldens come from the reference
        # built by g++ -ffp-contract=off (multiply, round, add): the two differ by at most one rounding of a
        # coordinate, 2^-52 relative -- eleven orders of magnitude inside the 1e-5 bar
        ulp = float(np.spacing(np.abs(g["adv_P"][:, :3]).max()))
        sc.cudaAdvect(float(g["dt"]))
        assert np.array_equal(sc.particles, g["adv_P"])
        assert np.array_equal(sc.vels, g["adv_vel"])
        assert np.abs(sc.disps - g["adv_disp"]).max() <= ulp and np.array_equal(sc.disps[:, 3], g["adv_disp"][:, 3])
        # ---- convexTetQuery: new cell, or -(start cell + 1) for a wall hit (ConvexQuery.cu:204-215)
        sc.convexTetQuery()
        ids, ref = sc.ids, g["loc_tet"]
        wall = ref < 0
        assert wall.sum() > 10 and np.array_equal(ids < 0, wall)
        assert np.array_equal(ids[~wall], ref[~wall] // 12)
        assert np.array_equal(-ids[wall] - 1, cell0[wall]) and np.array_equal((-ref[wall] - 1) // 12, cell0[wall])
        assert np.array_equal(sc.particles, g["adv_P"])               # the locator does not move P
        # ---- convexWallReflect: P = P_hit, disp = P_end - P_hit, vel mirrored, id = final cell
        sc.convexWallReflect()
        assert np.array_equal(sc.ids, g["ref_tet"] // 12)
        assert _rel(sc.particles, g["ref_P"], diag).max() <= REL_TOL
        assert _rel(sc.disps, g["ref_disp"], diag).max() <= REL_TOL
        assert np.abs(sc.vels[:, :3] - g["ref_vel"][:, :3]).max() <= 1e-9 * np.abs(g["ref_vel"][:, :3]).max()
        assert np.array_equal(sc.particles[~wall], g["ref_P"][~wall])  # untouched particles: bit-exact positions
        assert np.abs(sc.disps[~wall] - g["ref_disp"][~wall]).max() <= ulp
        # ---- cudaMoveParticles: P += disp, disp.xyz = 0
        sc.cudaMoveParticles()
        assert _rel(sc.particles, g["mov_P"], diag).max() <= REL_TOL
        assert np.abs(sc.particles[~wall] - g["mov_P"][~wall]).max() <= 2 * ulp
        assert np.array_equal(sc.disps, g["mov_disp"])
        assert _rel(sc.particles, g["mov_P"], diag).max() < 1e-12
    finally:
        sc.close()


def test_seeding_vs_reference_golden(gpu_ctx_factory, pitz):
    """cpf_seed_box / cpf_stage_seed_box (order = 1: g++'s argument evaluation order, what oracle/_ref did) against
    the reference's cudaInitParticles LCG<16> stream, bit for bit (cuda/particles.cu:78-97)."""
    from cudaparticlesfoam_amd.api import StagedCloud
    g = np.load(os.path.join(G, "init_particles.npz"))
    n = g["P"].shape[0]
    ctx = gpu_ctx_factory()
    ctx.set_mesh(pitz["mesh"]); ctx.set_velocity(pitz["U_uniform"])
    ctx.seed_box(n, g["lower"], g["upper"], 1)
    xyzw, _ = ctx.get_particles()
    assert np.array_equal(xyzw, g["P"])
    sc = StagedCloud(ctx, n)
    try:
        sc.cudaInitParticles(g["lower"], g["upper"], 1)
        assert np.array_equal(sc.particles, g["P"])
    finally:
        sc.close()
    # order = 0 (left-to-right evaluation, what nvcc might do) is the same stream with x and z exchanged
    ctx.seed_box(n, g["lower"], g["upper"], 0)
    sw, _ = ctx.get_particles()
    r = (g["P"][:, :3] - g["lower"]) / (g["upper"] - g["lower"])
    r0 = (sw[:, :3] - g["lower"]) / (g["upper"] - g["lower"])
    assert np.allclose(r0[:, 0], r[:, 2], atol=1e-12) and np.allclose(r0[:, 2], r[:, 0], atol=1e-12)
    assert np.allclose(r0[:, 1], r[:, 1], atol=1e-12)


def test_vertex_velocity_vs_reference_golden(gpu_ctx_factory, oracle_libs):
    """The reference's "VertexVelocity" advect mode (cudaAdvect(..., "VertexVelocity") -> particleAdvectKernel,
    cuda/particles.cu:244-313, 428-437) through cpf_set_tets / cpf_set_vertex_velocity / cpf_stage_advect_vertex and the
    other three stage calls, 60 cycles with wall reflections, against the reference's own output: cells identical,
    positions within 1e-5 of the domain diagonal for every particle; the advect stage bit-identical to its CPU statement."""
    from cudaparticlesfoam_amd.api import StagedCloud
    from cudaparticlesfoam_amd.cases import box_mesh
    from oracle.tetmesh import poly_to_tets
    g = np.load(os.path.join(G, "vertex_box.npz"))
    mesh = box_mesh(10, 9, 8)
    lo, hi = mesh.bounds(); diag = float(np.linalg.norm(hi - lo))
    pos, tets, tcell, _ = poly_to_tets(mesh, None, np.zeros((mesh.n_cells, 3)))
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh)
    ctx.set_tets(pos, tets, 12)
    ctx.set_vertex_velocity(g["vertex_U"])
    n = g["xyz0"].shape[0]
    cell0 = (g["tet0"] // 12).astype(np.int32)
    P0 = np.ones((n, 4)); P0[:, :3] = g["xyz0"]
    cw = oracle_libs.CellWalk()
    sc = StagedCloud(ctx, n)
    try:
        sc.set(P0, cell0)
        done = 0
        for k in g["checkpoints"]:
            for c in range(int(k) - done):
                if done == 0 and c == 0:
                    Pc = P0.copy(); vc = np.zeros((n, 4)); dc = np.zeros((n, 4))
                    cw.advect_vertex(Pc, cell0, vc, dc, float(g["dt"]), tets, 12, pos, g["vertex_U"])
                sc.cudaAdvect(float(g["dt"]), "VertexVelocity")
                if done == 0 and c == 0:
                    assert np.array_equal(sc.vels, vc) and np.array_equal(sc.disps, dc)          # HIP == CPU statement
                    assert np.abs(sc.vels - g["adv_vel"]).max() < 1e-13                           # == the reference
                sc.convexTetQuery(); sc.convexWallReflect(); sc.cudaMoveParticles()
            done = int(k)
            assert np.array_equal(sc.ids, g["tet_%d" % k] // 12), "k=%d" % k
            # (an interpolated field has velocity gradients: a rounding difference where a particle sits on a face
            # shared by two tets grows along the trajectory -- still orders of magnitude inside the bar)
            assert _rel(sc.particles, g["P_%d" % k], diag).max() <= REL_TOL
        with pytest.raises(ValueError):
            sc.cudaAdvect(0.1, "NoSuchVelocityMode")               # (the reference silently does nothing for a typo, cuda/particles.cu:417-445)
    finally:
        sc.close()


def test_fused_vertex_velocity_cycle_equals_the_staged_calls(gpu_ctx_factory):
    """cpf_step(..., CPF_STEP_VERTEX_VELOCITY): the reference's cycle with the "VertexVelocity" advect (cuda/particles.cu:
    244-313, 428-437 -> ConvexQuery -> reflect -> move) as ONE launch of the generic walk.  Same stages, same arithmetic as the
    five staged calls: cells and positions equal them bit for bit at every checkpoint of the reference golden -- single-cycle
    launches, fused launches and a sorted cloud alike -- and therefore the reference's own output within the same 1e-5."""
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.api import StagedCloud
    from cudaparticlesfoam_amd.cases import box_mesh
    from oracle.tetmesh import poly_to_tets
    g = np.load(os.path.join(G, "vertex_box.npz"))
    mesh = box_mesh(10, 9, 8)
    lo, hi = mesh.bounds(); diag = float(np.linalg.norm(hi - lo))
    pos, tets, tcell, _ = poly_to_tets(mesh, None, np.zeros((mesh.n_cells, 3)))
    n = g["xyz0"].shape[0]
    cell0 = (g["tet0"] // 12).astype(np.int32)
    dt = float(g["dt"])

    def ctx_with_field():
        c = gpu_ctx_factory()
        c.set_mesh(mesh); c.set_velocity(np.zeros((mesh.n_cells, 3)))
        c.set_tets(pos, tets, 12); c.set_vertex_velocity(g["vertex_U"])
        return c
    staged_ctx = ctx_with_field()
    P0 = np.ones((n, 4)); P0[:, :3] = g["xyz0"]
    sc = StagedCloud(staged_ctx, n)
    fused = []
    for mode in ("per cycle", "fused launches", "sorted", "generic walk", "generic walk, all tets"):
        c = ctx_with_field()
        if mode.startswith("generic walk"):
            c.set_option("step_variant", 0)                    # step_kernel_vertex: one thread per particle, CSR walk
            c.set_option("vertex_fast", 0 if mode.endswith("all tets") else 1)
        c.set_particles(g["xyz0"], cell0)
        if mode == "sorted":
            c.sort_by_cell()
        fused.append((mode, c))
    # since round 6 the cycle streams (all-hex mesh, a tet fan per cell): the streaming kernel with the interpolated advect
    assert fused[0][1].step_kernel_name(0.0, L.STEP_VERTEX_VELOCITY) == "cpf::step_kernel_stream_vertex<false, true, false, true, 1> (cone locate)"
    assert fused[3][1].step_kernel_name(0.0, L.STEP_VERTEX_VELOCITY) == "cpf::step_kernel_vertex<false, true, false> (cone locate)"
    assert fused[4][1].step_kernel_name(0.0, L.STEP_VERTEX_VELOCITY) == "cpf::step_kernel_vertex<false, true, false> (all tets)"
    try:
        sc.set(P0, cell0)
        done = 0
        for k in g["checkpoints"]:
            for _ in range(int(k) - done):
                sc.cudaAdvect(dt, "VertexVelocity"); sc.convexTetQuery(); sc.convexWallReflect(); sc.cudaMoveParticles()
            for mode, c in fused:
                fl = L.STEP_VERTEX_VELOCITY | (L.STEP_FUSE_CYCLES if mode in ("fused launches", "generic walk") else 0)
                c.step(dt, 0.0, int(k) - done, fl)
                xyzw, cell = c.get_particles()
                assert np.array_equal(cell, sc.ids), (mode, int(k))
                assert np.array_equal(xyzw[:, :3], sc.particles[:, :3]), (mode, int(k))
                assert np.array_equal(cell, g["tet_%d" % k] // 12) and _rel(np.c_[xyzw[:, :3], np.ones(n)], g["P_%d" % k], diag).max() <= REL_TOL
            done = int(k)
        # without the tets / vertex field the flag is refused, not ignored
        bare = gpu_ctx_factory()
        bare.set_mesh(mesh); bare.set_velocity(np.zeros((mesh.n_cells, 3))); bare.set_particles(g["xyz0"], cell0)
        with pytest.raises(L.CpfError):
            bare.step(dt, 0.0, 1, L.STEP_VERTEX_VELOCITY)
    finally:
        sc.close()


# ---- the oracle .so files built ON THIS BOX, tied to the same goldens in the same run ----------------------
@pytest.mark.parametrize("name", ["pitz_uniform", "pitz_analytic", "box_random"])
def test_oracle_built_here_reproduces_goldens(name, pitz, oracle_libs):
    og.test_tetwalk_reproduces_reference_goldens_bitwise(name, pitz, oracle_libs)
    og.test_cellwalk_matches_reference_goldens(name, pitz, oracle_libs)


def test_oracle_built_here_stage_face_seed_goldens(oracle_libs):
    og.test_stage_by_stage_goldens(oracle_libs)
    og.test_face_table_golden(oracle_libs)
    og.test_seeding_golden(oracle_libs)
    og.test_vertex_velocity_golden(oracle_libs)


def test_constant_velocity_advect_mode(gpu_ctx_factory):
    """cudaAdvect(..., "ConstantVelocity") (cuda/particles.cu:439-445 -> particleAdvectConstVel :376-399): every particle
    keeps the velocity already stored for it, disp = (vel * dt, -1); negative id -> w = 0 and nothing else; w == 0 ->
    untouched.  One rounding per component, so the comparison with numpy is exact."""
    from cudaparticlesfoam_amd.api import StagedCloud
    from cudaparticlesfoam_amd.cases import box_mesh
    ctx = gpu_ctx_factory()
    ctx.set_mesh(box_mesh(2, 2, 2))
    n = 5000
    rng = np.random.default_rng(4)
    P = np.concatenate([rng.uniform(0, 2, size=(n, 3)), np.ones((n, 1))], 1)
    P[::7, 3] = 0.0                                            # already switched off
    ids = rng.integers(0, 8, size=n).astype(np.int32)
    ids[::5] = -3                                              # left the domain
    vels = np.concatenate([rng.normal(size=(n, 3)), -np.ones((n, 1))], 1)
    sc = StagedCloud(ctx, n)
    sc.set(P, ids); sc._put("vels", vels); sc._put("disps", np.full((n, 4), 7.0))
    dt = 0.0123
    sc.cudaAdvect(dt, "ConstantVelocity")
    ctx.synchronize()
    Pn, d = sc.particles, sc.disps
    live = (P[:, 3] != 0) & (ids >= 0)
    assert np.array_equal(d[live, :3], vels[live, :3] * dt) and np.all(d[live, 3] == -1.0)
    assert np.all(d[~live] == 7.0)                              # not touched
    assert np.array_equal(Pn[:, :3], P[:, :3]) and np.array_equal(sc.vels, vels)
    assert np.array_equal(Pn[:, 3], np.where((P[:, 3] != 0) & (ids < 0), 0.0, P[:, 3]))
    with pytest.raises(ValueError):
        sc.cudaAdvect(dt, "NoSuchMode")
    sc.close()
