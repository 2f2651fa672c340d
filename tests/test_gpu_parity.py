"""GPU parity tests proper: the HIP path, called through the C-ABI, against the CPU oracles.

Bars: cells and counters bit-exact; positions bit-exact against the cell-walk statement
(same formulation); against the reference algorithm (tet walk, oracle/tetwalk.c == the
reference's own functions, see test_oracle_vs_ref.py) the tolerance BASELINE.json states:
|dx| / L <= 1e-5 (L = domain bbox diagonal) -- and we demand it for EVERY particle here.
"""
import numpy as np
import pytest

from conftest import domain_diag

pytestmark = pytest.mark.gpu
REL_TOL = 1e-5


def _seed_points(pz, n, box, seed=12345):
    return pz.uniform_points(seed, n, *box)


def _has_variant(ctx, variant):
    """Step variants 1, 2 and 5 (measured slower on every mesh) are only in EXPERIMENTS=1 builds of the library: the
    default build refuses them (cpf_set_option -> CPF_ERR_ARG)."""
    from cudaparticlesfoam_amd import _lib as L
    try:
        ctx.set_option("step_variant", variant)
        return True
    except L.CpfError as e:
        assert e.status == L.CPF_ERR_ARG and variant in (1, 2, 5) and "EXPERIMENTS=1" in str(e)
        return False


@pytest.fixture(scope="module")
def setup(pitz, oracle_libs, gpu_ctx_factory):
    cw = oracle_libs.CellWalk()
    tw = oracle_libs.TetWalk()
    mesh = pitz["mesh"]
    tables = cw.build(mesh)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh)
    return dict(cw=cw, tw=tw, mesh=mesh, tables=tables, ctx=ctx, pz=pitz["pz"], pitz=pitz)


def test_mesh_tables_match_oracle_build(setup):
    off, planes, nbr = setup["ctx"].mesh_tables()
    t = setup["tables"]
    assert np.array_equal(off, t.cell_off)
    assert np.array_equal(nbr, t.nbr)
    assert np.array_equal(planes, t.planes)          # bit-exact plane coefficients
    info = setup["ctx"].mesh_info()
    assert info["n_cells"] == 12225 and info["n_slots"] == 49180 + 24170


def test_mesh_layer_recognises_what_the_walk_shortcuts_need(setup, gpu_ctx_factory):
    """pitzDaily must come out all-hex, z-layered and one cell thick -- the z-pair skip and the mirrored end point of the
    Brownian kick depend on it, silently (for most of round 3 rounding noise in the face normals kept both switched off);
    a 3-D box is all-hex and neither; a refined pitzDaily is z-layered with mixed records."""
    from cudaparticlesfoam_amd.cases import box_mesh, refined_pitzdaily
    assert setup["ctx"].mesh_flags() == dict(all_hex=1, z_layered=1, z_thin=1, mixed=0)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(box_mesh(5, 4, 3))
    assert ctx.mesh_flags() == dict(all_hex=1, z_layered=1, z_thin=0, mixed=0)       # three layers: z faces exact, not all on the boundary
    ctx.set_mesh(box_mesh(5, 4, 1))
    assert ctx.mesh_flags() == dict(all_hex=1, z_layered=1, z_thin=1, mixed=0)
    ctx.set_option("z_fold", 0)
    assert ctx.mesh_flags()["z_thin"] == 0
    ctx.set_option("z_fold", 1)
    ctx.set_mesh(refined_pitzdaily()[0])
    assert ctx.mesh_flags() == dict(all_hex=0, z_layered=1, z_thin=1, mixed=1)


def test_lookup_method_follows_the_particles_per_cell(setup):
    """Which instantiation of the streaming kernel a step launches (cpf_step_kernel_name): sparse clouds (< 8 particles per
    cell) the pipelined-gather one, up to 128 per cell the fixed tag compare, above that the loop over distinct cells."""
    ctx, pz = setup["ctx"], setup["pz"]
    ctx.set_option("step_variant", -1); ctx.set_option("stream_lookup", -1)
    ctx.set_velocity(setup["pitz"]["U_uniform"])
    # (139 per cell: the loop lookup -- with the FLAT walk where it applies: a 2-D mesh, a field without a z component, no kick)
    for n, want in ((20_000, ", 4>"), (300_000, ", 9>"), (1_700_000, ", 8>")):                 # 12 225 cells: 1.6, 24.5, 139 per cell
        ctx.set_particles(_seed_points(pz, n, pz.DOMAIN_BOX, seed=3))
        assert ctx.step_kernel_name(0.0, 0).endswith(want), (n, ctx.step_kernel_name(0.0, 0))
    assert ctx.step_kernel_name(1e-5, 0).endswith(", 0>")                                       # the kick moves particles in z
    ctx.set_option("flat_walk", 0)
    assert ctx.step_kernel_name(0.0, 0).endswith(", 0>")
    ctx.set_option("flat_walk", 1)
    Uz = setup["pitz"]["U_uniform"].copy(); Uz[77, 2] = 1e-300
    ctx.set_velocity(Uz)                                                                        # ONE cell with a z component
    assert ctx.step_kernel_name(0.0, 0).endswith(", 0>")
    # a field handed over as a device array: the note "no z component" comes back asynchronously -- flat once it has been
    # seen to arrive, never before
    import torch
    Ud = torch.from_numpy(np.ascontiguousarray(setup["pitz"]["U_uniform"])).to("cuda:0")
    torch.cuda.synchronize()
    ctx.set_velocity_dev(Ud.data_ptr(), Ud.shape[0])
    ctx.synchronize()
    assert ctx.step_kernel_name(0.0, 0).endswith(", 8>")
    Uzd = torch.from_numpy(Uz).to("cuda:0")
    torch.cuda.synchronize()
    ctx.set_velocity_dev(Uzd.data_ptr(), Uzd.shape[0])
    assert ctx.step_kernel_name(0.0, 0).endswith(", 0>")                                        # (pending or arrived: not flat)
    ctx.synchronize()
    assert ctx.step_kernel_name(0.0, 0).endswith(", 0>")
    ctx.set_velocity(setup["pitz"]["U_uniform"])


@pytest.mark.parametrize("n,want", [(1_600_000, ", 8>"), (300_000, ", 9>")])
@pytest.mark.parametrize("field", ["U_uniform", "U_analytic"])
def test_flat_walk_bit_exact(setup, field, n, want):
    """The FLAT instantiation (csrc/cpf_walk.h "flat walk": 2-D mesh extruded straight in z, no z velocity, no kick -- the headline
    configuration): four side faces with two-term dot products, no z pair, no z in the walk.  Bit for bit (compared as BITS: the
    argument is about signs of zeros) the CPU statement and the same library with ``flat_walk`` 0 -- plain and fused launches,
    statistics, stored velocities, particles with z == -0.0 and on the front / back planes."""
    from cudaparticlesfoam_amd import _lib as L
    pz, ctx, cw, t = setup["pz"], setup["ctx"], setup["cw"], setup["tables"]
    U = setup["pitz"][field]
    xyz = _seed_points(pz, n, pz.DOMAIN_BOX, seed=41)             # (> 128 per cell: the loop lookup; fewer: the fixed compare)
    zlo, zhi = pz.DOMAIN_BOX[0][2], pz.DOMAIN_BOX[1][2]
    xyz[::7, 2] = zlo; xyz[1::7, 2] = zhi                           # on the front / back planes
    if zlo < 0 < zhi:
        xyz[2::7, 2] = -0.0
    dt = 4e-4 if field == "U_uniform" else 2e-4
    ctx.set_option("step_variant", -1); ctx.set_option("stream_lookup", -1)
    x, y, z = (xyz[:, k].copy() for k in range(3))
    c = cw.locate_initial(x, y, z, t, nthreads=cw.max_threads)
    st = np.zeros(3, np.int64)
    for k in (1, 3, 8):
        st += np.asarray(cw.step(x, y, z, c, dt, k, t, U, nthreads=cw.max_threads))
    outs = {}
    for flat in (1, 0):
        ctx.set_option("flat_walk", flat)
        ctx.set_option("stats", 1)
        ctx.set_velocity(U); ctx.set_particles(xyz); ctx.locate_initial(); ctx.sort_by_cell()
        c0 = ctx.counters()
        ctx.step(dt, 0.0, 1, 0); ctx.step(dt, 0.0, 3, L.STEP_FUSE_CYCLES); ctx.step(dt, 0.0, 8, L.STEP_STORE_VEL)
        c1 = ctx.counters()
        assert ctx.step_kernel_name(0.0, 0).endswith(want if flat else {", 8>": ", 0>", ", 9>": ", 1>"}[want])
        xyzw, cell, vel = ctx.get_particles(want_vel=True)
        outs[flat] = (xyzw[:, :3].copy(), cell.copy(), vel[:, :3].copy())
        assert c1["cells_visited"] - c0["cells_visited"] == int(st[0]) and c1["reflections"] - c0["reflections"] == int(st[1])
    ctx.set_option("flat_walk", 1)
    # (the cloud was sorted: compare through the sort's own order -- both runs sort alike)
    assert np.array_equal(outs[1][1], outs[0][1])
    assert np.array_equal(outs[1][0].view(np.int64), outs[0][0].view(np.int64))
    live = outs[1][1] >= 0                                          # (a particle outside the mesh never gets a velocity stored)
    assert np.array_equal(outs[1][2][live].view(np.int64), outs[0][2][live].view(np.int64))
    key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]   # noqa: E731
    ref = np.stack([x, y, z], 1)
    assert np.array_equal(key(outs[1][0]).view(np.int64), key(ref).view(np.int64))
    assert np.array_equal(np.sort(outs[1][1]), np.sort(c))


@pytest.mark.parametrize("how", ["last_cell_z", "nan_z", "negative_zero_z", "dev_then_set_stream"])
def test_flat_walk_is_decided_by_the_whole_field(how, setup):
    """The flat walk runs only for a field WITHOUT a z component, found by the kernel that lays the field out (round-4 advisory):
    a z component in the very last cell, or a NaN one, switches it off (the particles there do move in z: equal to the CPU
    statement); -0.0 everywhere keeps it on; and after cpf_set_velocity_dev -- the note arrives asynchronously -- a change of
    stream before the step neither loses nor misreads it."""
    import torch
    pz, ctx, cw, t, mesh = setup["pz"], setup["ctx"], setup["cw"], setup["tables"], setup["mesh"]
    U = setup["pitz"]["U_analytic"].copy()
    ctx.set_option("step_variant", -1); ctx.set_option("stream_lookup", -1); ctx.set_option("flat_walk", 1)
    n = 2_000_000
    xyz = _seed_points(pz, n, pz.DOMAIN_BOX, seed=43)
    flat = True
    if how == "last_cell_z":
        U[-1, 2] = 0.37; flat = False
    elif how == "nan_z":
        U[mesh.n_cells // 2, 2] = np.nan; flat = False
    elif how == "negative_zero_z":
        U[:, 2] = -0.0
    dev = torch.device("cuda", 0)
    if how == "dev_then_set_stream":
        s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        ctx.set_stream(s1.cuda_stream)
        dU = torch.from_numpy(U).to(dev); torch.cuda.synchronize()
        ctx.set_velocity_dev(dU.data_ptr(), mesh.n_cells)       # (the "no z component" note is read back behind it, asynchronously)
        ctx.set_stream(s2.cuda_stream)                           # ... and the stream changes before anybody has seen it
    else:
        ctx.set_velocity(U)
    ctx.set_particles(xyz); ctx.locate_initial(); ctx.sort_by_cell()
    ctx.step(2e-4, 0.0, 6, 0)
    name = ctx.step_kernel_name(0.0, 0)
    assert name.endswith(", 8>" if flat else ", 0>"), name
    got, gc = ctx.get_particles()
    if how != "nan_z":                                           # (a NaN velocity has no CPU answer worth comparing)
        x, y, z = (xyz[:, k].copy() for k in range(3))
        c = cw.locate_initial(x, y, z, t, nthreads=cw.max_threads)
        cw.step(x, y, z, c, 2e-4, 6, t, U, nthreads=cw.max_threads)
        assert np.array_equal(gc, c) and np.array_equal(got[:, 0], x) and np.array_equal(got[:, 1], y) and np.array_equal(got[:, 2], z)
        if how == "last_cell_z":
            assert (np.abs(z - xyz[:, 2]) > 0).any()             # somebody did move in z
    ctx.use_own_stream()
    ctx.set_velocity(setup["pitz"]["U_analytic"])


def test_initial_locate_matches_bruteforce(setup):
    pz, ctx, cw, t = setup["pz"], setup["ctx"], setup["cw"], setup["tables"]
    n = 20000
    lo, hi = np.array(pz.DOMAIN_BOX[0]), np.array(pz.DOMAIN_BOX[1])
    xyz = _seed_points(pz, n, (lo - 0.002, hi + 0.002), seed=7)    # some points outside the mesh
    ctx.set_particles(xyz)
    n_out = ctx.locate_initial()
    _, cell = ctx.get_particles()
    ref = cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t, nthreads=cw.max_threads)
    assert np.array_equal(cell, ref)
    assert n_out == int((ref < 0).sum()) and 0 < n_out < n


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("field", ["U_uniform", "U_analytic"])
def test_step_bit_exact_vs_cellwalk(setup, field, variant):
    """Every kernel variant (generic CSR walk, all-hex fixed-slot walk, + wave-uniform scalar plane fetches,
    wave-cooperative LDS cell cache) against the CPU statement, bit for bit, sorted and unsorted order."""
    pz, ctx, cw, t = setup["pz"], setup["ctx"], setup["cw"], setup["tables"]
    U = setup["pitz"][field]
    n = 100000
    xyz = _seed_points(pz, n, pz.DOMAIN_BOX, seed=99)
    if not _has_variant(ctx, variant):
        pytest.skip("step variant %d is an experiment: not in the default build (make EXPERIMENTS=1)" % variant)
    ctx.set_velocity(U)
    ctx.set_particles(xyz)
    ctx.locate_initial()
    _, cell0 = ctx.get_particles()
    if variant >= 2:
        ctx.sort_by_cell()                 # the scalar path only triggers on cell-coherent waves
    x, y, z = (xyz[:, k].copy() for k in range(3))
    c = cell0.copy()
    done = 0
    c0 = ctx.counters()
    hops = refl = 0
    for k in (1, 9, 90):
        ctx.step(1e-4, 0.0, k)
        st = cw.step(x, y, z, c, 1e-4, k, t, U, nthreads=cw.max_threads)
        hops += int(st[0]); refl += int(st[1])
        done += k
        xyzw, cell = ctx.get_particles()
        assert np.array_equal(cell, c), "cells differ after %d cycles" % done
        assert np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) and np.array_equal(xyzw[:, 2], z), \
            "positions not bit-identical after %d cycles" % done
    c1 = ctx.counters()
    assert c1["cells_visited"] - c0["cells_visited"] == hops
    assert c1["reflections"] - c0["reflections"] == refl
    assert refl > 0   # the case does exercise wall reflection
    ctx.set_option("step_variant", -1)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("field", ["U_uniform", "U_analytic"])
def test_production_instantiation_bit_exact(setup, field, fused):
    """The kernel a host gets by default -- statistics OFF, the 7-waves-per-SIMD instantiation of the
    wave-cooperative walk -- against the CPU statement, bit for bit (the fixture's contexts switch the
    statistics on, which selects a different instantiation of the same source)."""
    from cudaparticlesfoam_amd import _lib as L
    pz, ctx, cw, t = setup["pz"], setup["ctx"], setup["cw"], setup["tables"]
    U = setup["pitz"][field]
    n = 300_001
    xyz = _seed_points(pz, n, pz.DOMAIN_BOX, seed=4242)
    ctx.set_option("stats", 0)
    try:
        ctx.set_velocity(U)
        ctx.set_particles(xyz)
        ctx.locate_initial()
        _, cell0 = ctx.get_particles()
        ctx.sort_by_cell()
        before = ctx.counters()
        ctx.step(1e-4, 0.0, 40, L.STEP_FUSE_CYCLES if fused else 0)
        xyzw, cell = ctx.get_particles()
        assert ctx.counters() == before                                  # nothing was counted
    finally:
        ctx.set_option("stats", 1)
    x, y, z = (xyz[:, k].copy() for k in range(3))
    c = cell0.copy()
    st = cw.step(x, y, z, c, 1e-4, 40, t, U, nthreads=cw.max_threads)
    assert int(st[1]) > 1000                                             # reflections were exercised
    assert np.array_equal(cell, c)
    assert np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) and np.array_equal(xyzw[:, 2], z)


@pytest.mark.parametrize("field", ["U_uniform", "U_analytic"])
def test_step_vs_reference_algorithm(setup, field, oracle_libs):
    """HIP cell walk vs the reference's tet walk on the 12-tets-per-cell decomposition."""
    from oracle.tetmesh import poly_to_tets
    pz, ctx, cw, tw, mesh = setup["pz"], setup["ctx"], setup["cw"], setup["tw"], setup["mesh"]
    U = setup["pitz"][field]
    pos, tets, tcell, tu = poly_to_tets(mesh, setup["pitz"]["centres"], U)
    m = tw.tables(pos, tets, tu)
    n = 4096
    xyz = _seed_points(pz, n, pz.INLET_BOX, seed=12345)
    ctx.set_velocity(U)
    ctx.set_particles(xyz)
    assert ctx.locate_initial() == 0
    _, cell0 = ctx.get_particles()
    P = np.zeros((n, 4)); P[:, :3] = xyz; P[:, 3] = 1
    ids = (cell0 * 12).astype(np.int32)
    tw.bary_query(P, ids, m)                       # reference initial-locate fix-up (query/RTQuery.cu:189-218)
    assert np.array_equal(ids // 12, cell0)
    vels = np.zeros((n, 4)); disps = np.zeros((n, 4))
    L = domain_diag(mesh)
    done = 0
    for k in (1, 9, 90, 900):
        ctx.step(1e-4, 0.0, k)
        tw.cycles(P, ids, vels, disps, 1e-4, k, m, nthreads=tw.max_threads)
        done += k
        xyzw, cell = ctx.get_particles()
        rel = np.linalg.norm(xyzw[:, :3] - P[:, :3], axis=1) / L
        assert rel.max() <= REL_TOL, "after %d cycles max rel err %.3e" % (done, rel.max())
        same = (ids // 12 == cell) | ((ids < 0) & (cell < 0))
        assert same.all(), "%d cell mismatches after %d cycles" % ((~same).sum(), done)
        assert np.array_equal(xyzw[:, 3] != 0, P[:, 3] != 0)


def test_fused_cycles_and_sort_do_not_change_results(setup, gpu_ctx_factory):
    from cudaparticlesfoam_amd import _lib as L
    pz, mesh = setup["pz"], setup["mesh"]
    U = setup["pitz"]["U_analytic"]
    xyz = _seed_points(pz, 50000, pz.DOMAIN_BOX, seed=5)
    outs = []
    for mode in ("plain", "fused", "sorted"):
        ctx = gpu_ctx_factory()
        ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz); ctx.locate_initial()
        if mode == "sorted":
            ctx.sort_by_cell()
        ctx.step(1e-4, 0.0, 40, L.STEP_FUSE_CYCLES if mode == "fused" else 0)
        if mode == "sorted":
            ctx.sort_by_cell()
        outs.append(ctx.get_particles())
        ctx.close()
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])


def test_kernel_choice_per_launch(setup):
    """Default step_variant -1: the streaming kernel, for single-cycle launches and fused ones of any length (until round 4
    launches that fuse eight or more cycles went to the wave-cooperative kernel, which was faster there then); an explicit
    variant is obeyed.  Results are the same either way."""
    from cudaparticlesfoam_amd import _lib as L
    pz, ctx = setup["pz"], setup["ctx"]
    ctx.set_velocity(setup["pitz"]["U_analytic"])
    xyz = _seed_points(pz, 30000, pz.DOMAIN_BOX, seed=12)
    got = []
    for variant in (-1, 4, 3):
        ctx.set_option("step_variant", variant)
        ctx.set_particles(xyz); ctx.locate_initial(); ctx.sort_by_cell()
        ctx.step(1e-4, 0.0, 2)
        name1 = ctx.step_kernel_name(0.0, 0)
        ctx.step(1e-4, 0.0, 2, L.STEP_FUSE_CYCLES)
        name2 = ctx.step_kernel_name(0.0, L.STEP_FUSE_CYCLES)
        ctx.step(1e-4, 0.0, 9, L.STEP_FUSE_CYCLES)
        name9 = ctx.step_kernel_name(0.0, L.STEP_FUSE_CYCLES)
        # (-1: the streaming kernel for fused launches of any length since round 4 -- it used to hand nine cycles to the other one)
        want = {-1: ("stream", "stream", "stream"), 4: ("stream", "stream", "stream"), 3: ("coop", "coop", "coop")}[variant]
        assert tuple("coop" if "step_kernel_coop" in nm else "stream" if "step_kernel_stream" in nm else nm
                     for nm in (name1, name2, name9)) == want
        got.append(ctx.get_particles())
    for g in got[1:]:
        assert np.array_equal(g[0], got[0][0]) and np.array_equal(g[1], got[0][1])
    ctx.set_option("step_variant", -1)


def test_coop_kernel_cell_limit_guard(setup):
    """The wave-cooperative kernel addresses cell records with 32-bit byte offsets (256 B x 2^24 cells).  Above that limit
    the streaming kernel runs instead, for an explicit step_variant 3 and for fused launches alike; the test hook
    coop_max_cells moves the limit below pitzDaily's 12 225 cells so that the switch can be seen, results unchanged."""
    from cudaparticlesfoam_amd import _lib as L
    pz, ctx = setup["pz"], setup["ctx"]
    ctx.set_velocity(setup["pitz"]["U_analytic"])
    xyz = _seed_points(pz, 20000, pz.DOMAIN_BOX, seed=77)
    got = []
    for limit in (0, 5000):
        ctx.set_option("coop_max_cells", limit)
        names = []
        for variant in (3, -1):
            ctx.set_option("step_variant", variant)
            ctx.set_particles(xyz); ctx.locate_initial(); ctx.sort_by_cell()
            ctx.step(1e-4, 0.0, 9, L.STEP_FUSE_CYCLES)
            names.append(ctx.step_kernel_name(0.0, L.STEP_FUSE_CYCLES))
            got.append(ctx.get_particles())
        # names[0]: the explicit variant 3; names[1]: the default, the streaming kernel whatever the limit
        assert ("step_kernel_coop" in names[0]) == (limit == 0) and ("step_kernel_stream" in names[0]) == (limit != 0), names
        assert "step_kernel_stream" in names[1], names
    for g in got[1:]:
        assert np.array_equal(g[0], got[0][0]) and np.array_equal(g[1], got[0][1])
    ctx.set_option("coop_max_cells", 0); ctx.set_option("step_variant", -1)


def test_box_uniform_flow_known_answers(oracle_libs, gpu_ctx_factory):
    """Uniform U in a box: P_k = P_0 + k*dt*U until the first wall; specular reflection at x = L."""
    from cudaparticlesfoam_amd.cases import box_mesh
    mesh = box_mesh(8, 6, 4, lower=(0, 0, 0), upper=(2.0, 1.5, 1.0))
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh)
    U = np.tile([0.37, 0.0, 0.0], (mesh.n_cells, 1))
    ctx.set_velocity(U)
    rng = np.random.default_rng(3)
    n = 5000
    xyz = rng.uniform([0.05, 0.05, 0.05], [1.0, 1.45, 0.95], size=(n, 3))
    ctx.set_particles(xyz)
    assert ctx.locate_initial() == 0
    dt, k = 0.05, 20                      # travel 0.37 < 1.0: no wall yet
    ctx.step(dt, 0.0, k)
    xyzw, cell = ctx.get_particles()
    assert np.allclose(xyzw[:, 0], xyz[:, 0] + k * dt * 0.37, rtol=0, atol=1e-13)
    assert np.array_equal(xyzw[:, 1], xyz[:, 1]) and np.array_equal(xyzw[:, 2], xyz[:, 2])
    # drive everything into the x = 2 wall; positions fold back: x' = 2L - x (modulo the per-step re-advect)
    ctx.step(dt, 0.0, 200)
    xyzw, cell = ctx.get_particles()
    assert (cell >= 0).all() and (xyzw[:, 0] <= 2.0 + 1e-12).all() and (xyzw[:, 0] >= 2.0 - 0.37 * dt - 1e-12).all()
    assert ctx.counters()["reflections"] > 0 and ctx.counters()["lost"] == 0


def test_box_random_field_vs_reference_algorithm(oracle_libs, gpu_ctx_factory):
    """Heavy-reflection case on a 3-D box with random cell-constant U (SURVEY.md fact 2 probe)."""
    from cudaparticlesfoam_amd.cases import box_mesh
    from oracle.tetmesh import poly_to_tets
    tw = oracle_libs.TetWalk()
    mesh = box_mesh(10, 9, 8)
    rng = np.random.default_rng(11)
    U = rng.normal(size=(mesh.n_cells, 3))
    pos, tets, tcell, tu = poly_to_tets(mesh, None, U)
    m = tw.tables(pos, tets, tu)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(U)
    n = 8000
    xyz = rng.uniform([0, 0, 0], [10, 9, 8], size=(n, 3))
    ctx.set_particles(xyz)
    assert ctx.locate_initial() == 0
    _, cell0 = ctx.get_particles()
    P = np.zeros((n, 4)); P[:, :3] = xyz; P[:, 3] = 1
    ids = (cell0 * 12).astype(np.int32)
    tw.bary_query(P, ids, m)
    vels = np.zeros((n, 4)); disps = np.zeros((n, 4))
    L = domain_diag(mesh)
    for k in (1, 19, 80):
        ctx.step(0.3, 0.0, k)
        tw.cycles(P, ids, vels, disps, 0.3, k, m, nthreads=tw.max_threads)
        xyzw, cell = ctx.get_particles()
        rel = np.linalg.norm(xyzw[:, :3] - P[:, :3], axis=1) / L
        bad = rel > REL_TOL
        # a particle that grazes a cell edge within rounding may take the neighbouring cell: count, don't hide
        assert bad.mean() <= 1e-4, "%d of %d beyond tolerance" % (bad.sum(), n)
    assert ctx.counters()["reflections"] > 1000


def test_inactive_and_no_reflect(setup, gpu_ctx_factory):
    from cudaparticlesfoam_amd import _lib as L
    pz, mesh = setup["pz"], setup["mesh"]
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(setup["pitz"]["U_uniform"])
    xyz = np.array([[0.5, 0.5, 0.5], [0.28, 0.0, 0.0], [-0.01, 0.01, 0.0]])   # outside, near outlet, inlet channel
    ctx.set_particles(xyz)
    assert ctx.locate_initial() == 1
    ctx.step(1e-4, 0.0, 1)
    xyzw, cell = ctx.get_particles()
    assert cell[0] == L.CELL_FROZEN and xyzw[0, 3] == 0.0 and np.array_equal(xyzw[0, :3], xyz[0])
    ctx.step(1e-4, 0.0, 30, L.STEP_NO_REFLECT)       # particle 1 reaches the outlet: lost, then frozen
    xyzw, cell = ctx.get_particles()
    assert cell[1] == L.CELL_FROZEN and cell[2] >= 0
    assert ctx.counters()["lost"] == 1


def test_brownian_statistics_and_stream(setup, gpu_ctx_factory):
    from cudaparticlesfoam_amd.cases import box_mesh
    cw = setup["cw"]
    mesh = box_mesh(4, 4, 4, lower=(-1, -1, -1), upper=(1, 1, 1))
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(np.zeros((mesh.n_cells, 3)))
    n = 200000
    ctx.set_particles(np.zeros((n, 3)) + 1e-3)
    ctx.locate_initial()
    D, dt = 1.5e-5, 1e-4
    ctx.set_seed(1234)
    ctx.step(dt, D, 1)
    xyzw, cell = ctx.get_particles()
    d = xyzw[:, :3] - 1e-3
    sigma = np.sqrt(2 * D * dt)
    assert np.abs(d.mean(0)).max() < 5 * sigma / np.sqrt(n)
    assert np.abs(d.var(0) / sigma ** 2 - 1).max() < 0.02
    # same counter-based stream as the CPU statement (libm in float vs the fp32 hardware log2 / sin / cos: 1e-6)
    worst = 0.0
    for g in list(range(0, 2000)) + [n - 1]:
        xi = cw.normal3(g, 0, 1234)
        worst = max(worst, float(np.abs(d[g] / sigma - xi).max()))
    assert worst < 2e-5, worst


def test_call_order_errors(gpu_ctx_factory, pitz):
    from cudaparticlesfoam_amd import _lib as L
    ctx = gpu_ctx_factory()
    with pytest.raises(L.CpfError) as e:
        ctx.set_velocity(np.zeros((10, 3)))
    assert e.value.status == L.CPF_ERR_STATE
    ctx.set_mesh(pitz["mesh"])
    with pytest.raises(L.CpfError) as e:
        ctx.set_velocity(np.zeros((10, 3)))
    assert e.value.status == L.CPF_ERR_ARG
    ctx.set_particles(np.zeros((4, 3)))
    with pytest.raises(L.CpfError) as e:
        ctx.step(1e-4, 0.0, 1)
    assert e.value.status == L.CPF_ERR_STATE
    bad = pitz["mesh"]
    import copy
    m2 = copy.copy(bad); m2.owner = bad.owner.copy(); m2.owner[5] = bad.n_cells + 3
    with pytest.raises(L.CpfError) as e:
        ctx.set_mesh(m2)
    assert e.value.status == L.CPF_ERR_MESH


def test_handoff_pack_unpack_single_gpu(setup, gpu_ctx_factory):
    """The multi-GPU hand-off kernels on one GPU: pretend to be rank 1 of 4, pack, then check the split
    against numpy -- every leaver is in the send buffer under its destination (index order), every stayer is
    still in [0, nStay), nothing is duplicated or lost; unpack appends the records back."""
    import torch
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.parallel import slab_cell_ranges
    pz, mesh, ctx = setup["pz"], setup["mesh"], setup["ctx"]
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)      # same stream as the torch ops that fill the arrays
    n = 300001
    xyz = _seed_points(pz, n, pz.DOMAIN_BOX, seed=21)
    x, y, z = (torch.from_numpy(xyz[:, k].copy()).to(dev) for k in range(3))
    cell = torch.empty(n, dtype=torch.int32, device=dev)
    gid = torch.arange(n, dtype=torch.int64, device=dev) + 5_000_000_000          # ids beyond 2^32 survive the trip
    p = lambda t: t.data_ptr()   # noqa: E731
    ctx.locate_initial_dev(p(x), p(y), p(z), p(cell), n)
    torch.cuda.synchronize()
    world, rank = 4, 1
    cell_lo = slab_cell_ranges(setup["pitz"]["vols"], world)
    lo_dev = torch.from_numpy(cell_lo.copy()).to(dev)
    cap = n
    sendbuf = torch.zeros(cap * L.HANDOFF_DOUBLES, dtype=torch.float64, device=dev)
    counts = torch.zeros(16, dtype=torch.int64, device=dev); nstay = torch.zeros(1, dtype=torch.int64, device=dev)
    h = dict(x=x.cpu().numpy(), y=y.cpu().numpy(), z=z.cpu().numpy(), c=cell.cpu().numpy(), g=gid.cpu().numpy())
    ctx.pack_leavers_dev(p(x), p(y), p(z), p(cell), p(gid), n, p(lo_dev), world, rank, p(sendbuf), cap, p(counts), p(nstay))
    ctx.synchronize(); torch.cuda.synchronize()
    owner = np.searchsorted(cell_lo[1:], h["c"], side="right")
    dest = np.where((h["c"] < 0) | (owner == rank), -1, owner)
    want_counts = [int((dest == r).sum()) for r in range(world)]
    assert counts[:world].cpu().tolist() == want_counts and want_counts[rank] == 0 and sum(want_counts) > 0
    ns = int(nstay.item())
    assert ns == int((dest < 0).sum())
    rec = sendbuf[: sum(want_counts) * L.HANDOFF_DOUBLES].cpu().numpy().reshape(-1, L.HANDOFF_DOUBLES)
    off = 0
    for r in range(world):
        idx = np.nonzero(dest == r)[0]                                             # index order within a destination
        blk = rec[off:off + idx.size]; off += idx.size
        assert np.array_equal(blk[:, 0], h["x"][idx]) and np.array_equal(blk[:, 1], h["y"][idx])
        assert np.array_equal(blk[:, 2], h["z"][idx]) and np.array_equal(blk[:, 3].astype(np.int32), h["c"][idx])
        assert np.array_equal(blk[:, 4].astype(np.int64), h["g"][idx])
    g_after = gid[:ns].cpu().numpy()
    stay_ids = h["g"][dest < 0]
    assert np.array_equal(np.sort(g_after), np.sort(stay_ids))                     # stayers: a permutation, none lost
    order = np.argsort(h["g"]); pos = order[np.searchsorted(h["g"][order], g_after)]
    assert np.array_equal(x[:ns].cpu().numpy(), h["x"][pos]) and np.array_equal(cell[:ns].cpu().numpy(), h["c"][pos])
    # unpack: append all records again -> the full multiset of particles is back
    ctx.unpack_arrivals_dev(p(x), p(y), p(z), p(cell), p(gid), ns, p(sendbuf), sum(want_counts))
    ctx.synchronize(); torch.cuda.synchronize()
    assert np.array_equal(np.sort(gid.cpu().numpy()), np.sort(h["g"]))
    back = gid.cpu().numpy(); pos = order[np.searchsorted(h["g"][order], back)]
    assert np.array_equal(z.cpu().numpy(), h["z"][pos]) and np.array_equal(cell.cpu().numpy(), h["c"][pos])
    ctx.use_own_stream()


@pytest.mark.parametrize("order", ["sorted", "shuffled"])
def test_cell_histogram_matches_bincount(setup, order):
    """Per-cell particle counts x scale (input of the ownership re-cut) == numpy.bincount, for cell-sorted input
    (one LDS atomic per run) and for shuffled input (per-lane atomics); lost/frozen states are ignored."""
    import torch
    mesh, ctx = setup["mesh"], setup["ctx"]
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(31)
    n = 1_000_003
    c = rng.integers(0, mesh.n_cells, size=n).astype(np.int32)
    c[rng.random(n) < 0.01] = -1
    c[rng.random(n) < 0.01] = -2
    c[:5000] = 17                                                                    # one hot cell
    if order == "sorted":
        c = np.sort(c)
    cell = torch.from_numpy(c).to(dev)
    w = torch.full((mesh.n_cells,), 77.0, dtype=torch.float64, device=dev)          # must be overwritten, not added to
    want = np.bincount(c[c >= 0], minlength=mesh.n_cells)
    for scale in (1.0, 0.375):
        ctx.cell_histogram_dev(cell.data_ptr(), n, scale, w.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(w.cpu().numpy(), want * scale)
    ctx.cell_histogram_dev(cell.data_ptr(), 0, 1.0, w.data_ptr())                   # empty shard -> all zero
    torch.cuda.synchronize()
    assert float(w.abs().sum().item()) == 0.0
    ctx.use_own_stream()


def test_cell_histogram_global_path_on_a_large_mesh(gpu_ctx_factory):
    """Meshes with more cells than fit an LDS histogram (> 32768) take the global-atomic path."""
    import torch
    from cudaparticlesfoam_amd.cases import box_mesh
    mesh = box_mesh(40, 40, 25)                                                      # 40 000 cells
    ctx = gpu_ctx_factory(); ctx.set_mesh(mesh)
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(2)
    c = np.sort(rng.integers(-2, mesh.n_cells, size=300_001).astype(np.int32))
    c = np.concatenate([c, rng.integers(0, mesh.n_cells, size=7777).astype(np.int32)])   # sorted body + unsorted tail
    w = torch.empty(mesh.n_cells, dtype=torch.float64, device=dev)
    ctx.cell_histogram_dev(torch.from_numpy(c).to(dev).data_ptr(), c.size, 2.0, w.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(w.cpu().numpy(), 2.0 * np.bincount(c[c >= 0], minlength=mesh.n_cells))


@pytest.mark.parametrize("n_ranks", [1, 2, 3, 8, 16])
def test_cell_ranges_kernel_matches_host_rule(setup, n_ranks):
    """Equal-weight cuts on the device == parallel.slab_cell_ranges on integer-valued weights (exact prefix
    sums), including empty cells, an empty cloud and all the weight in one cell."""
    import torch
    from cudaparticlesfoam_amd.parallel import slab_cell_ranges
    mesh, ctx = setup["mesh"], setup["ctx"]
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(n_ranks)
    cases = [rng.integers(0, 2000, size=mesh.n_cells).astype(np.float64),
             (rng.integers(0, 50, size=mesh.n_cells) * (rng.random(mesh.n_cells) < 0.05)).astype(np.float64),
             np.zeros(mesh.n_cells), np.eye(1, mesh.n_cells, 4321)[0] * 9.0]
    lo = torch.full((n_ranks + 1,), -7, dtype=torch.int32, device=dev)
    for w in cases:
        ctx.cell_ranges_dev(torch.from_numpy(w).to(dev).data_ptr(), n_ranks, lo.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(lo.cpu().numpy(), slab_cell_ranges(w, n_ranks))
    ctx.use_own_stream()


def test_deferred_handoff_catch_up_is_bit_exact(setup, gpu_ctx_factory):
    """The overlapped hand-off's device side on one GPU, call by call (the *_dev entry points the shard layer strings
    together): split the cloud as rank 0 of 2, keep stepping the OLD range for 3 cycles (the stale tail must be inert), then
    append the 'arrivals' (the send buffer itself) and let them replay the 3 cycles in one fused launch.  With Brownian
    motion on, every particle must end exactly where plain stepping puts it: the (gid, step) Philox streams make the replay
    order-independent."""
    import torch
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.parallel import slab_cell_ranges
    pz, mesh, cw, t = setup["pz"], setup["mesh"], setup["cw"], setup["tables"]
    U = setup["pitz"]["U_analytic"]
    dev = torch.device("cuda", 0)
    ctx = gpu_ctx_factory(); ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_seed(99)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 150_000
    xyz = _seed_points(pz, n, pz.DOMAIN_BOX, seed=5)
    x, y, z = (xyz[:, k].copy() for k in range(3))
    c = cw.locate_initial(x, y, z, t)
    gid = np.arange(n, dtype=np.int64) + 10
    cell_lo = torch.from_numpy(slab_cell_ranges(setup["pitz"]["vols"], 2)).to(dev)
    p = lambda a: a.data_ptr()     # noqa: E731

    def upload():
        return [torch.from_numpy(a).to(dev) for a in (x, y, z, c, gid)]
    dx, dy, dz, dc, dg = upload()
    sendbuf = torch.empty(n * L.HANDOFF_DOUBLES, dtype=torch.float64, device=dev)
    counts = torch.zeros(2, dtype=torch.int64, device=dev); nstay = torch.zeros(1, dtype=torch.int64, device=dev)
    dt, D = 1e-4, 1e-6
    ctx.step_dev(p(dx), p(dy), p(dz), p(dc), p(dg), None, n, dt, D, 0, 2, 0)                     # steps 0, 1
    ctx.pack_leavers_dev(p(dx), p(dy), p(dz), p(dc), p(dg), n, p(cell_lo), 2, 0, p(sendbuf), n, p(counts), p(nstay))   # split after step 1
    cnt = counts.cpu().tolist(); n_stay = int(nstay.item())
    assert cnt[0] == 0 and cnt[1] > 1000 and n_stay + cnt[1] == n
    assert bool((dc[n_stay:n] == L.CELL_LOST).all())                       # stale tail marked inert
    ctx.step_dev(p(dx), p(dy), p(dz), p(dc), p(dg), None, n, dt, D, 2, 3, 0)                     # steps 2, 3, 4 on the OLD range
    ctx.unpack_arrivals_dev(p(dx), p(dy), p(dz), p(dc), p(dg), n_stay, p(sendbuf), cnt[1])       # arrivals = what was sent
    ctx.step_dev(p(dx) + 8 * n_stay, p(dy) + 8 * n_stay, p(dz) + 8 * n_stay, p(dc) + 4 * n_stay, p(dg) + 8 * n_stay, None, cnt[1],
                 dt, D, 2, 3, L.STEP_FUSE_CYCLES)                                                # they replay steps 2..4
    torch.cuda.synchronize()
    g, gx, gy, gz, gc = (a.cpu().numpy() for a in (dg, dx, dy, dz, dc))
    # the answer: the same kernel stepping the undisturbed cloud 5 cycles (device log/cos differ from libm in the
    # last bits, so the Brownian term is compared GPU to GPU; the D = 0 walk is pinned to the oracle elsewhere)
    qx, qy, qz, qc, qg = upload()
    ctx.step_dev(p(qx), p(qy), p(qz), p(qc), p(qg), None, n, dt, D, 0, 5, 0)
    torch.cuda.synchronize()
    px, py, pz_, pc = (a.cpu().numpy() for a in (qx, qy, qz, qc))
    o = g - 10
    assert np.array_equal(np.sort(o), np.arange(n))
    assert np.array_equal(gx, px[o]) and np.array_equal(gy, py[o]) and np.array_equal(gz, pz_[o])
    assert np.array_equal(gc, pc[o])
    moved = np.abs(px - x).max()
    assert moved > 1e-4                                                       # and the cloud did move


@pytest.mark.parametrize("overlap", [0, 3, -1])
def test_sharded_cloud_one_rank_rccl_group(setup, gpu_ctx_factory, overlap):
    """The N>1 path behind the C-ABI (cpf_shard_*: histogram -> ncclAllReduce -> device re-cut -> split -> ncclAllGather of
    the counts -> grouped ncclSend/ncclRecv all-to-all-v on the side stream -> unpack -> catch-up) on ONE GPU with a real
    one-rank RCCL communicator made by the library (cpf_comm_create): particle results must not depend on it."""
    from cudaparticlesfoam_amd import _lib as L
    import torch
    from cudaparticlesfoam_amd.parallel import Communicator, ShardedCloud, unique_id
    pz, mesh, cw, t = setup["pz"], setup["mesh"], setup["cw"], setup["tables"]
    U = setup["pitz"]["U_analytic"]
    dev = torch.device("cuda", 0)
    comm = Communicator(unique_id(L.COMM_RCCL), 0, 1, 0)
    try:
        ctx = gpu_ctx_factory()
        ctx.set_mesh(mesh)
        ctx.set_velocity(U)
        n = 200_000
        xyz = _seed_points(pz, n, pz.DOMAIN_BOX, seed=77)
        x, y, z = (xyz[:, k].copy() for k in range(3))
        c = cw.locate_initial(x, y, z, t)
        cloud = ShardedCloud(ctx, [0, mesh.n_cells], n + 64, comm, send_fraction=1.0, exchange_interval=0)
        cloud.force_collectives = True
        cloud.rebalance_interval = 3
        cloud.sort_interval = 4
        cloud.overlap_steps = overlap                   # side-stream counts + all-to-all while the loop runs on
        cloud.enable_time_balancing()
        tx, ty, tz = (torch.from_numpy(a).to(dev) for a in (x, y, z))
        torch.cuda.synchronize()
        cloud.set_particles(tx, ty, tz, None, None)
        cloud.step(1e-4, 10)
        assert cloud.rebalances == 3 and cloud.n == n and list(cloud.cell_lo) == [0, mesh.n_cells]
        g, gx, gy, gz, gc = cloud.gather_to_numpy()
        cw.step(x, y, z, c, 1e-4, 10, t, U)
        assert np.array_equal(np.sort(g), np.arange(n))
        assert np.array_equal(gx, x[g]) and np.array_equal(gy, y[g]) and np.array_equal(gz, z[g])
        assert np.array_equal(gc, c[g])
        # the whole cloud in particle-id order through the collective gather (one rank: the same particles)
        xyzw, cell, _ = cloud.gather(0)
        assert np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) and np.array_equal(xyzw[:, 2], z)
        assert np.array_equal(cell, c)
        cloud.close()
    finally:
        comm.close()


def test_full_size_properties(setup, gpu_ctx_factory):
    """BASELINE.json's full size (1e7 particles, pitzDaily) through size-independent properties: particle
    count conserved, every active particle lies inside the cell it claims (all plane distances <= 1e-9 on a
    1e5 sample), before the first wall uniform flow is exactly linear, FUSE_CYCLES == per-cycle launches."""
    import torch
    import bench
    from cudaparticlesfoam_amd import _lib as L
    pz, mesh, ctx = setup["pz"], setup["mesh"], setup["ctx"]
    dev = torch.device("cuda", 0)
    n = 10_000_000
    ctx.set_velocity(setup["pitz"]["U_uniform"])
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 77, dev)
    x0 = x.clone(); y0 = y.clone()
    p = lambda t: t.data_ptr()   # noqa: E731
    x2, y2, z2, c2 = x.clone(), y.clone(), z.clone(), c.clone()
    k = 12
    ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, 1e-4, 0.0, 0, k, 0)
    ctx.step_dev(p(x2), p(y2), p(z2), p(c2), None, None, n, 1e-4, 0.0, 0, k, L.STEP_FUSE_CYCLES)
    torch.cuda.synchronize()
    assert torch.equal(x, x2) and torch.equal(y, y2) and torch.equal(z, z2) and torch.equal(c, c2)
    assert int((c >= 0).sum()) == n                                    # all boundaries reflect: nobody is lost
    # linearity where no wall can have been reached: start x < 0.25 m, 12 mm of travel, channel interior
    far = (x0 < 0.19) & (y0.abs() < 0.012) & (x0 > 0.0)
    dx = (x - x0)[far]
    assert float((dx - k * 1e-4 * 10.0).abs().max()) < 1e-13 and float((y - y0)[far].abs().max()) == 0.0
    # inside-own-cell check on a sample, with the host-built planes
    off, planes, nbr = ctx.mesh_tables()
    idx = torch.randint(0, n, (100000,), device=dev)
    xs, ys, zs, cs = (t[idx].cpu().numpy() for t in (x, y, z, c))
    pl = planes.reshape(-1, 6, 4)[cs]
    fd = pl[:, :, 3] - (pl[:, :, 0] * xs[:, None] + pl[:, :, 1] * ys[:, None] + pl[:, :, 2] * zs[:, None])
    assert fd.max() <= 1e-9
    ctx.use_own_stream()


def test_generic_polyhedral_path_on_reference_tet_mesh(oracle_libs, gpu_ctx_factory):
    """Cells == the tets of the reference's own test geometry (createBoxMesh): 4-faced cells force the generic
    CSR walk.  HIP == cell-walk statement bit for bit, and == the reference tet walk (same elements, per-tet
    velocity, ~1e5 wall reflections) to rounding, every particle."""
    from tetcells import box_tets, tet_cell_polymesh
    tw, cw = oracle_libs.TetWalk(), oracle_libs.CellWalk()
    pos, tets = box_tets(6, 5, 4)
    mesh = tet_cell_polymesh(pos, tets)
    rng = np.random.default_rng(4)
    U = rng.normal(size=(mesh.n_cells, 3))
    m = tw.tables(pos, tets, U); t = cw.build(mesh)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(U)
    n = 20000
    xyz = rng.uniform([0, 0, 0], [6, 5, 4], size=(n, 3))
    ctx.set_particles(xyz)
    assert ctx.locate_initial() == 0
    _, cell0 = ctx.get_particles()
    assert np.array_equal(cell0, cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t,
                                                   nthreads=cw.max_threads))
    P = np.zeros((n, 4)); P[:, :3] = xyz; P[:, 3] = 1
    ids = cell0.copy(); vels = np.zeros((n, 4)); disps = np.zeros((n, 4))
    x, y, z, c = xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), cell0.copy()
    for k in (1, 9, 50):
        ctx.step(0.2, 0.0, k)
        cw.step(x, y, z, c, 0.2, k, t, U, nthreads=cw.max_threads)
        tw.cycles(P, ids, vels, disps, 0.2, k, m, nthreads=tw.max_threads)
        xyzw, cell = ctx.get_particles()
        assert np.array_equal(cell, c) and np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) \
            and np.array_equal(xyzw[:, 2], z)
        assert np.array_equal(cell, ids)
        assert np.abs(xyzw[:, :3] - P[:, :3]).max() <= 1e-11
    assert ctx.counters()["reflections"] > 50000


def test_config2_one_million_particles_bit_exact(setup):
    """BASELINE.json configs[1]: pitzDaily frozen U, 1e6 particles, positions checked against the CPU statement
    after 1, 10 and 100 cycles (bit for bit) -- and therefore, through the oracle chain, within rounding of the
    reference arithmetic (tests/test_oracle_golden.py: <= 1e-14 relative after 1000 cycles)."""
    pz, ctx, cw, t = setup["pz"], setup["ctx"], setup["cw"], setup["tables"]
    U = setup["pitz"]["U_analytic"]
    n = 1_000_000
    xyz = _seed_points(pz, n, pz.INLET_BOX, seed=12345)          # SURVEY.md 8d config 2 seeding
    ctx.set_velocity(U)
    ctx.set_particles(xyz)
    assert ctx.locate_initial() == 0
    _, cell0 = ctx.get_particles()
    ctx.sort_by_cell()
    x, y, z, c = xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), cell0.copy()
    for k in (1, 9, 90):
        ctx.step(1e-4, 0.0, k)
        cw.step(x, y, z, c, 1e-4, k, t, U, nthreads=cw.max_threads)
        xyzw, cell = ctx.get_particles()
        assert np.array_equal(cell, c)
        assert np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) and np.array_equal(xyzw[:, 2], z)


def test_tutorial_case_end_to_end(tmp_path, pitz):
    """BASELINE.json configs[0] shape on the GPU path: the tutorial dictionary (cudaParticlesDict:17-29, inverted
    seeding box, D = 1.5e-5, dt 1e-4, deltaT 0.1 => 1000 cycles in ONE advect.H pass), 1e4 particles, through the
    host mirror of the fragments with the library's VTU writer."""
    import os
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.api import CudaParticles
    pz = pitz["pz"]
    d = dict(pz.PARTICLE_DICT, numParticles=10000, saveInterval=250)
    frames = []

    def writer(frame, xyzw, vel, cell):
        frames.append((frame, xyzw.copy(), cell.copy()))

    p = CudaParticles(pitz["mesh"], pitz["U_analytic"], d, writer=writer)
    p.ctx.set_option("stats", 1)                                  # diagnostics are off by default
    assert p.outOfDomain == 0 and frames[0][0] == 0
    assert p.advect(100.0, 0.1) == 0                              # before startTime 282: gate closed (advect.H:33)
    assert p.advect(300.0, 0.1) == 1000
    assert [f[0] for f in frames] == [0, 1, 251, 501, 751]        # step % saveInterval == 0 -> frame step+1
    xyzw, cell = p.particles()
    assert (cell >= 0).all() and (xyzw[:, 3] == 1).all()          # every boundary reflects: nobody leaves
    lo, hi = pitz["mesh"].bounds()
    assert (xyzw[:, :3] >= lo - 1e-12).all() and (xyzw[:, :3] <= hi + 1e-12).all()
    moved = np.linalg.norm(xyzw[:, :3] - frames[0][1][:, :3], axis=1)
    assert moved.mean() > 0.02                                    # 0.1 s at O(1-10) m/s
    c = p.ctx.counters()
    assert c["particle_steps"] == 10000 * 1000 and c["lost"] == 0
    lib = L.load()
    ke = C_double()
    path = str(tmp_path / "particle_1000.vtu").encode()
    import ctypes
    assert lib.cpf_write_vtu(p.ctx.h, path, ctypes.byref(ke)) == 0 and os.path.getsize(path) > 10000 * 40
    p.close()


def C_double():
    import ctypes
    return ctypes.c_double()


def test_config5_transient_velocity_on_refined_mesh(oracle_libs, gpu_ctx_factory, pitz):
    """BASELINE.json configs[4] shape (pimple-like): a 16x larger mesh (195 600 hex cells, TJunction scale),
    U re-uploaded before every Eulerian step as nCells x 3 doubles (the reference re-uploads 12 copies per
    cell, src/advect.H:44-57), sub-cycled like advect.H.  HIP == CPU statement bit for bit at every step."""
    import time
    pz = pitz["pz"]
    cw = oracle_libs.CellWalk()
    mesh = pz.pitzdaily_mesh(refine=4)
    assert mesh.n_cells == 16 * 12225
    centres, _ = mesh.cell_centres_volumes()
    t = cw.build(mesh)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh)
    base = pz.analytic_step_u(mesh, centres)
    n = 200000
    xyz = _seed_points(pz, n, pz.DOMAIN_BOX, seed=8)
    ctx.set_velocity(base)
    ctx.set_particles(xyz)
    ctx.locate_initial()
    _, cell0 = ctx.get_particles()
    assert np.array_equal(cell0, cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t,
                                                   nthreads=cw.max_threads))
    keep = cell0 >= 0
    ctx.set_particles(xyz[keep], cell0[keep])
    ctx.sort_by_cell()
    x, y, z, c = xyz[keep, 0].copy(), xyz[keep, 1].copy(), xyz[keep, 2].copy(), cell0[keep].copy()
    upload = 0.0
    for step in range(5):
        U = base * (1.0 + 0.3 * np.sin(0.7 * step)) + np.array([0.0, 0.4 * np.cos(step), 0.0])   # transient field
        t0 = time.perf_counter(); ctx.set_velocity(U); upload += time.perf_counter() - t0
        ctx.step(2.5e-5, 0.0, 8)                    # finer mesh => shorter Lagrangian dt (50-cell walk cap)
        cw.step(x, y, z, c, 2.5e-5, 8, t, U, nthreads=cw.max_threads)
        xyzw, cell = ctx.get_particles()
        assert np.array_equal(cell, c)
        assert np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) and np.array_equal(xyzw[:, 2], z)
    assert upload / 5 < 0.05                        # 4.7 MB per refresh; the reference would move 56 MB


@pytest.mark.parametrize("lookup", [0, 1, 4])
@pytest.mark.parametrize("sort", [True, False])
def test_stream_record_lookup_methods(setup, lookup, sort):
    """The streaming kernel finds a wave's cells in its record cache either by a loop over the distinct cells or by a
    fixed compare against the tag vector, the latter also in its form for sparse clouds (pipelined per-lane gathers;
    option stream_lookup; picked per launch from the particles per cell).  All three, on a sorted cloud (1-3 cells per wave) and an unsorted one (up to 64: gather rounds, lanes sitting rounds out),
    with and without the Brownian kick: bit-identical to the CPU statement."""
    pz, ctx, cw, t = setup["pz"], setup["ctx"], setup["cw"], setup["tables"]
    U = setup["pitz"]["U_analytic"]
    n = 60000
    xyz = _seed_points(pz, n, pz.DOMAIN_BOX, seed=4242)
    ctx.set_option("step_variant", 4)
    ctx.set_option("stream_lookup", lookup)
    ctx.set_option("sort_interval", 0 if not sort else 50)
    ctx.set_velocity(U)
    for D in (0.0, 1.5e-5):
        ctx.set_particles(xyz)
        ctx.locate_initial()
        _, cell0 = ctx.get_particles()
        if sort:
            ctx.sort_by_cell()
        x, y, z = (xyz[:, k].copy() for k in range(3))
        c = cell0.copy()
        name = ctx.step_kernel_name(D)
        assert name.endswith(", %d>" % lookup) and "step_kernel_stream" in name
        if D == 0.0:
            for k in (1, 12):
                ctx.step(1e-4, D, k)
                cw.step(x, y, z, c, 1e-4, k, t, U, nthreads=cw.max_threads)
                xyzw, cell = ctx.get_particles()
                assert np.array_equal(cell, c) and np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y)
        else:
            # same (seed, gid, step) stream for both methods: explicit step numbers at the device-array level
            import torch
            dev = torch.device("cuda", 0)
            xyzw, cell = ctx.get_particles()
            order = np.argsort(cell, kind="stable") if sort else np.arange(n)
            tx, ty, tz = (torch.from_numpy(xyzw[order, k].copy()).to(dev) for k in range(3))
            tc = torch.from_numpy(cell[order].copy()).to(dev)
            tg = torch.from_numpy(order.astype(np.int64)).to(dev)
            torch.cuda.synchronize()
            ctx.step_dev(tx.data_ptr(), ty.data_ptr(), tz.data_ptr(), tc.data_ptr(), tg.data_ptr(), None, n, 1e-4, D, 1000, 8, 0)
            ctx.synchronize()
            got = tuple(a.cpu().numpy() for a in (tx, ty, tz, tc))
            ref = setup.setdefault("_lookup_ref", {})
            if sort in ref:                                       # the methods agree bit for bit with each other
                assert all(np.array_equal(a, b) for a, b in zip(ref[sort], got))
            else:
                ref[sort] = got
    ctx.set_option("stream_lookup", -1)
    ctx.set_option("sort_interval", 50)
    ctx.set_option("step_variant", -1)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 1000])
def test_ragged_sizes(setup, n):
    """Cloud sizes around the wave (64) and block (256) boundaries, every kernel variant: no lane of a partial
    wave may write or read out of bounds, results stay bit-identical to the CPU statement."""
    pz, ctx, cw, t = setup["pz"], setup["ctx"], setup["cw"], setup["tables"]
    U = setup["pitz"]["U_analytic"]
    xyz = _seed_points(pz, n, pz.DOMAIN_BOX, seed=1000 + n)
    ctx.set_velocity(U)
    ref_c = cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t)
    x, y, z, c = xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), ref_c.copy()
    cw.step(x, y, z, c, 1e-4, 25, t, U)
    for variant in (0, 1, 2, 3, 4, 5):
        if not _has_variant(ctx, variant):
            continue
        ctx.set_particles(xyz)
        ctx.locate_initial()
        ctx.sort_by_cell()
        ctx.step(1e-4, 0.0, 25)
        xyzw, cell = ctx.get_particles()
        assert np.array_equal(cell, c) and np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y)
    ctx.set_option("step_variant", -1)


def test_all_particles_outside_the_mesh(setup):
    ctx = setup["ctx"]
    xyz = np.tile([[5.0, 5.0, 5.0]], (300, 1)) + np.arange(300)[:, None]
    ctx.set_particles(xyz)
    assert ctx.locate_initial() == 300
    before = ctx.counters()
    ctx.step(1e-4, 0.0, 3)
    xyzw, cell = ctx.get_particles()
    assert (cell == -2).all() and (xyzw[:, 3] == 0).all() and np.array_equal(xyzw[:, :3], xyz)   # frozen in place
    assert ctx.counters()["particle_steps"] == before["particle_steps"]


def test_auto_sort_is_invisible(setup, gpu_ctx_factory):
    """cpf_step re-sorts the owned cloud every `sort_interval` cycles; ids and stored velocities travel with the
    particles, so results (positions, cells, velocities by particle id) do not depend on the interval."""
    from cudaparticlesfoam_amd import _lib as L
    pz, mesh = setup["pz"], setup["mesh"]
    U = setup["pitz"]["U_analytic"]
    xyz = _seed_points(pz, 30000, pz.DOMAIN_BOX, seed=3)
    outs = []
    for interval in (0, 7, 25):
        ctx = gpu_ctx_factory()
        ctx.set_option("sort_interval", interval)
        ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz); ctx.locate_initial()
        for _ in range(6):
            ctx.step(1e-4, 0.0, 9)
            ctx.step(1e-4, 0.0, 1, L.STEP_STORE_VEL)
        outs.append(ctx.get_particles(want_vel=True))
        ctx.close()
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert np.array_equal(a, b)
    assert np.abs(outs[0][2][:, :3]).max() > 1.0          # velocities really are stored


@pytest.mark.parametrize("seed", [0, 3, 5, 9, 11])
def test_random_sheared_meshes_bit_exact(seed, oracle_libs, gpu_ctx_factory):
    """Random graded, sheared hexahedral blocks with random cell-constant U and time steps that cross several cells
    and bounce off several walls per step (tests/test_oracle_random.py shows these equal the reference algorithm)."""
    from test_oracle_random import _case
    cw = oracle_libs.CellWalk()
    rng, mesh, U, dt = _case(seed)
    t = cw.build(mesh)
    lo, hi = mesh.bounds()
    xyz = rng.uniform(lo, hi, size=(20000, 3))
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz)
    n_out = ctx.locate_initial()
    _, cell0 = ctx.get_particles()
    ref0 = cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t, nthreads=cw.max_threads)
    assert np.array_equal(cell0, ref0) and n_out == int((ref0 < 0).sum())
    x, y, z, c = xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), ref0.copy()
    for k in (1, 7, 40):
        ctx.step(dt, 0.0, k)
        cw.step(x, y, z, c, dt, k, t, U, nthreads=cw.max_threads)
        xyzw, cell = ctx.get_particles()
        assert np.array_equal(cell, c)
        assert np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) and np.array_equal(xyzw[:, 2], z)
    assert ctx.counters()["reflections"] > 1000
    ctx.close()


def test_sort_orders_by_cell_then_position(setup):
    """cpf_sort_by_cell_dev: cells non-decreasing (lost/frozen at the tail), a permutation (ids intact), and inside
    one cell the particles come in sub-box order.  pitzDaily is one cell thick in z and longest in x, so the key
    layout chosen at mesh ingest is 4 bins in x (leading) x 32 bins in y, none in z."""
    import torch
    pz, ctx, mesh = setup["pz"], setup["ctx"], setup["mesh"]
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 400000
    xyz = _seed_points(pz, n, (np.array(pz.DOMAIN_BOX[0]) - 0.003, np.array(pz.DOMAIN_BOX[1]) + 0.003), seed=17)
    x, y, z = (torch.from_numpy(xyz[:, k].copy()).to(dev) for k in range(3))
    c = torch.empty(n, dtype=torch.int32, device=dev); g = torch.arange(n, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    ctx.locate_initial_dev(p(x), p(y), p(z), p(c), n)
    ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
    torch.cuda.synchronize()
    cs, gs = c.cpu().numpy(), g.cpu().numpy()
    inside = cs >= 0
    k = int(inside.sum())
    assert inside[:k].all() and not inside[k:].any() and 0 < k < n           # outside particles at the tail
    assert (np.diff(cs[:k]) >= 0).all()
    assert np.array_equal(np.sort(gs), np.arange(n))
    assert np.array_equal(x.cpu().numpy(), xyz[gs, 0]) and np.array_equal(z.cpu().numpy(), xyz[gs, 2])
    # sub-cell order: recompute the bins on the host for the fullest cell
    big = np.bincount(cs[:k]).argmax()
    sel = cs == big
    pts = np.stack([x.cpu().numpy()[sel], y.cpu().numpy()[sel], z.cpu().numpy()[sel]], 1)
    off, cf = mesh.cell_faces(); fo = mesh.face_offsets
    vid = np.unique(np.concatenate([mesh.face_verts[fo[f]:fo[f + 1]] for f in cf[off[big]:off[big + 1]]]))
    lo, hi = mesh.points[vid].min(0), mesh.points[vid].max(0)
    nb = np.array([4.0, 32.0, 1.0])
    u = ((pts.astype(np.float32) - lo.astype(np.float32)) * (nb / (hi - lo)).astype(np.float32)).astype(np.int64)
    u = np.clip(u, 0, nb.astype(np.int64) - 1)
    sub = (u[:, 0] << 5) | u[:, 1]
    assert (np.diff(sub) >= 0).mean() > 0.999                                   # fp32 bin edges: allow a stray particle
    ctx.use_own_stream()


def test_out_of_place_sort_equals_in_place(setup):
    """cpf_sort_by_cell_dev_to writes the order cpf_sort_by_cell_dev produces into a second set of arrays and leaves its
    inputs alone; n = 0 and n = 1 included; aliasing is refused."""
    import torch
    from cudaparticlesfoam_amd._lib import CpfError
    pz, ctx = setup["pz"], setup["ctx"]
    dev = torch.device("cuda", 0)
    for n in (0, 1, 70001):
        xyz = _seed_points(pz, max(n, 1), pz.DOMAIN_BOX, seed=31 + n)[:n]
        x, y, z = (torch.from_numpy(xyz[:, k].copy()).to(dev) for k in range(3))
        c = torch.empty(n, dtype=torch.int32, device=dev)
        g = torch.arange(n, dtype=torch.int64, device=dev)
        p = lambda t: t.data_ptr() if t.numel() else 0   # noqa: E731
        if n:
            ctx.locate_initial_dev(p(x), p(y), p(z), p(c), n)
        torch.cuda.synchronize()
        keep = [t.clone() for t in (x, y, z, c, g)]
        out = [torch.zeros_like(t) for t in (x, y, z, c, g)]
        ctx.sort_by_cell_dev_to(p(x), p(y), p(z), p(c), p(g), *(p(t) for t in out), n)
        ctx.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(keep, (x, y, z, c, g)))          # inputs untouched
        ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
        ctx.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(out, (x, y, z, c, g)))
        if n > 1:
            with pytest.raises(CpfError):
                ctx.sort_by_cell_dev_to(p(x), p(y), p(z), p(c), p(g), p(x), p(out[1]), p(out[2]), p(out[3]), p(out[4]), n)


def test_64bit_labels_give_the_same_tables(setup, gpu_ctx_factory):
    """cpf_set_mesh_l64 (OpenFOAM built with WM_LABEL_SIZE=64) == cpf_set_mesh."""
    import copy
    mesh = setup["mesh"]
    m64 = copy.copy(mesh)
    m64.face_offsets = mesh.face_offsets.astype(np.int64); m64.face_verts = mesh.face_verts.astype(np.int64)
    m64.owner = mesh.owner.astype(np.int64); m64.neighbour = mesh.neighbour.astype(np.int64)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(m64)
    a = ctx.mesh_tables(); b = setup["ctx"].mesh_tables()
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    ctx.close()


def test_async_vtu_frame_is_a_snapshot(setup, gpu_ctx_factory, tmp_path):
    """cpf_write_vtu_async: the frame holds the cloud as it was at the call (the worker thread formats and writes
    while the GPU steps on), the kinetic energy is known at once, and the bytes are those of the synchronous
    writer; a second call waits for the first frame, cpf_write_vtu_wait for the last."""
    pz, mesh = setup["pz"], setup["mesh"]
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(setup["pitz"]["U_analytic"])
    n = 120_000
    ctx.set_particles(_seed_points(pz, n, pz.DOMAIN_BOX, seed=808))
    ctx.locate_initial()
    from cudaparticlesfoam_amd import _lib as L
    ctx.step(1e-4, 0.0, 3, L.STEP_STORE_VEL)
    ke_sync = ctx.write_vtu(tmp_path / "sync.vtu")
    ke_async = ctx.write_vtu_async(tmp_path / "async_a.vtu")
    ctx.step(1e-4, 0.0, 25, L.STEP_STORE_VEL)                       # the cloud moves on while frame a is written
    ke_b = ctx.write_vtu_async(tmp_path / "async_b.vtu")            # waits for frame a, snapshots again
    ctx.write_vtu_wait()
    a = open(tmp_path / "async_a.vtu", "rb").read()
    assert a == open(tmp_path / "sync.vtu", "rb").read() and ke_async == ke_sync and len(a) > n * 100
    b = open(tmp_path / "async_b.vtu", "rb").read()
    assert b != a and b.endswith(b"</VTKFile>\n") and ke_b != ke_sync
    ctx.write_vtu_wait()                                             # idempotent
    # without the energy the call returns after the device-side snapshot (one kernel launch, no wait for PCIe): the frame is the
    # cloud as it was at the call all the same -- the cycles queued right behind it do not leak into it
    ctx.write_vtu(tmp_path / "sync_c.vtu")
    assert ctx.write_vtu_async(tmp_path / "async_c.vtu", want_ke=False) is None
    ctx.step(1e-4, 0.0, 7, L.STEP_STORE_VEL)
    ctx.write_vtu_wait()
    assert open(tmp_path / "async_c.vtu", "rb").read() == open(tmp_path / "sync_c.vtu", "rb").read()
    # option "vtu_binary": the same frames with raw appended arrays -- what the cloud holds, exactly
    from test_vtu_writer import _read_appended
    ctx.set_option("vtu_binary", 1)
    ke_bin = ctx.write_vtu_async(tmp_path / "bin.vtu")
    ctx.write_vtu_wait()
    xyzw, cell = ctx.get_particles()
    d = _read_appended(tmp_path / "bin.vtu")
    assert ke_bin != ke_b and ke_bin > 0 and np.array_equal(d["Position"], xyzw[:, :3]) and np.array_equal(d["ConvexTetID"], cell)
    assert np.array_equal(d["ParticleType"], xyzw[:, 3].astype(np.int32)) and (np.abs(d["vels"]).max(0)[:2] > 0).all()
    ctx.write_vtu(tmp_path / "bin_sync.vtu")
    assert open(tmp_path / "bin_sync.vtu", "rb").read() == open(tmp_path / "bin.vtu", "rb").read()


@pytest.mark.parametrize("n_parts", [2, 5])
def test_rank_direct_ingest_equals_whole_mesh_ingest(setup, gpu_ctx_factory, n_parts):
    """cpf_set_mesh_parts on the pieces of the decomposed pitzDaily mesh builds the same walk tables as cpf_set_mesh
    on the whole mesh (planes bit for bit, same slots; only the boundary faces' codes differ, they are numbered in
    piece order), and the particles step identically."""
    from cudaparticlesfoam_amd.cases import split_into_parts
    pz, mesh, whole = setup["pz"], setup["mesh"], setup["ctx"]
    U = setup["pitz"]["U_analytic"]
    ctx = gpu_ctx_factory()
    ctx.set_mesh_parts(split_into_parts(mesh, n_parts))
    off_a, planes_a, nbr_a = whole.mesh_tables()
    off_b, planes_b, nbr_b = ctx.mesh_tables()
    assert np.array_equal(off_a, off_b) and np.array_equal(planes_a, planes_b)
    assert np.array_equal(nbr_a >= 0, nbr_b >= 0) and np.array_equal(nbr_a[nbr_a >= 0], nbr_b[nbr_b >= 0])
    n = 100_000
    xyz = _seed_points(pz, n, pz.DOMAIN_BOX, seed=606)
    out = []
    for c in (whole, ctx):
        c.set_velocity(U)                       # the ranks' U slices, concatenated in rank order, ARE the global U
        c.set_particles(xyz)
        c.locate_initial()
        c.step(1e-4, 0.0, 30)
        out.append(c.get_particles())
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])


def test_empty_shard_entry_points(setup, gpu_ctx_factory):
    """A rank whose cell range holds no particle calls every *_dev entry point with n = 0: nothing may be launched
    with an empty grid, nothing written, no error -- and the shard layer on top of them steps, sorts, re-cuts and
    exchanges an empty shard."""
    import torch
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.parallel import ShardedCloud
    mesh = setup["mesh"]
    dev = torch.device("cuda", 0)
    ctx = gpu_ctx_factory(); ctx.set_mesh(mesh); ctx.set_velocity(setup["pitz"]["U_uniform"])
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    cap, sentinel = 1024, 123.5
    x = torch.full((cap,), sentinel, dtype=torch.float64, device=dev); y = x.clone(); z = x.clone()
    cell = torch.full((cap,), 7, dtype=torch.int32, device=dev); gid = torch.zeros(cap, dtype=torch.int64, device=dev)
    alt = [torch.empty_like(a) for a in (x, y, z, cell, gid)]
    cell_lo = torch.tensor([0, 100, mesh.n_cells], dtype=torch.int32, device=dev)
    sendbuf = torch.empty(cap * L.HANDOFF_DOUBLES, dtype=torch.float64, device=dev)
    counts = torch.full((2,), 5, dtype=torch.int64, device=dev); nstay = torch.full((1,), 5, dtype=torch.int64, device=dev)
    weights = torch.ones(mesh.n_cells, dtype=torch.float64, device=dev)
    p = lambda a: a.data_ptr()     # noqa: E731
    ctx.step_dev(p(x), p(y), p(z), p(cell), p(gid), None, 0, 1e-4, 0.0, 0, 3, 0)
    ctx.step_dev(p(x), p(y), p(z), p(cell), p(gid), None, 0, 1e-4, 1e-6, 3, 2, L.STEP_FUSE_CYCLES)
    ctx.sort_by_cell_dev_to(p(x), p(y), p(z), p(cell), p(gid), *[p(a) for a in alt], 0)
    ctx.locate_initial_dev(p(x), p(y), p(z), p(cell), 0)
    ctx.pack_leavers_dev(p(x), p(y), p(z), p(cell), p(gid), 0, p(cell_lo), 2, 0, p(sendbuf), cap, p(counts), p(nstay))
    ctx.cell_histogram_dev(p(cell), 0, 1.0, p(weights))
    ctx.cell_ranges_dev(p(weights), 2, p(cell_lo))
    ctx.unpack_arrivals_dev(p(x), p(y), p(z), p(cell), p(gid), 0, p(sendbuf), 0)
    torch.cuda.synchronize()
    assert int(nstay.item()) == 0 and counts.tolist() == [0, 0]
    assert float(weights.abs().sum().item()) == 0.0
    assert cell_lo.tolist() == [0, 0, mesh.n_cells]                       # no weight anywhere: all cuts at 0
    assert bool((x == sentinel).all()) and bool((cell == 7).all())
    cloud = ShardedCloud(ctx, None, cap, None, send_fraction=1.0, exchange_interval=2)
    cloud.force_collectives = True
    cloud.rebalance_interval = 3; cloud.sort_interval = 2; cloud.overlap_steps = 1
    cloud.step(1e-4, 7)
    cloud.flush()
    assert cloud.n == 0 and cloud.global_count() == 0 and cloud.exchanges >= 3
    cloud.close()


def test_merge_failure_is_reported_through_the_context(setup, gpu_ctx_factory):
    """cpf_set_mesh_parts with a broken piece: CPF_ERR_MESH and the stitcher's reason in cpf_last_error(ctx)."""
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.cases import split_into_parts
    parts = split_into_parts(setup["mesh"], 2)
    parts[1].owner = parts[1].owner.copy(); parts[1].owner[3] = 10 ** 6
    ctx = gpu_ctx_factory()
    with pytest.raises(L.CpfError) as e:
        ctx.set_mesh_parts(parts)
    assert e.value.status == L.CPF_ERR_MESH
    msg = ctx.lib.cpf_last_error(ctx.h).decode()
    assert "piece 1" in msg and "owner" in msg


@pytest.mark.parametrize("variant", [4, 3, 0])
def test_stored_velocity_of_particles_lost_during_a_fused_launch(setup, gpu_ctx_factory, variant):
    """Walls that do not reflect (reflectWall = false, src/initCuda.H:67): a particle that reaches one is lost and frozen from the
    next cycle on, and its stored velocity stays what its LAST live cycle gave it (the reference's advect skips it from then on,
    cuda/particles.cu:333-338).  That must not depend on how the cycles are grouped into launches: 12 cycles as 12 launches, as
    one fused launch and as 5 + 7 give the same velocities -- round 4 found the streaming kernel skipping the store for particles
    lost in the middle of a fused launch (tools/stream_check.py)."""
    import torch
    from cudaparticlesfoam_amd import _lib as L
    mesh, pz = setup["mesh"], setup["pz"]
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(setup["pitz"]["U_uniform"]); ctx.set_option("step_variant", variant)
    n = 60_000
    xyz = pz.uniform_points(5, int(n * 1.5), (0.25, -0.025, -0.0005), (0.29, 0.025, 0.0005))     # the last 40 mm before the outlet
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    x0, y0, z0 = (torch.from_numpy(xyz[:, k].copy()).to(dev) for k in range(3))
    c0 = torch.empty(xyz.shape[0], dtype=torch.int32, device=dev)
    ctx.locate_initial_dev(x0.data_ptr(), y0.data_ptr(), z0.data_ptr(), c0.data_ptr(), xyz.shape[0])
    keep = (c0 >= 0).nonzero().flatten()[:n]
    x0, y0, z0, c0 = x0[keep].contiguous(), y0[keep].contiguous(), z0[keep].contiguous(), c0[keep].contiguous()
    n = int(x0.numel())
    fl = L.STEP_NO_REFLECT | L.STEP_STORE_VEL
    outs = []
    for groups in ([1] * 12, [12], [5, 7]):
        x, y, z, c = x0.clone(), y0.clone(), z0.clone(), c0.clone()
        vel = torch.full((3 * n,), -7.0, dtype=torch.float64, device=dev)
        step = 0
        for k in groups:
            ctx.step_dev(x.data_ptr(), y.data_ptr(), z.data_ptr(), c.data_ptr(), None, vel.data_ptr(), n, 2e-4, 0.0, step, k,
                         fl | (L.STEP_FUSE_CYCLES if k > 1 else 0))
            step += k
        torch.cuda.synchronize()
        outs.append((x.cpu().numpy(), c.cpu().numpy(), vel.cpu().numpy().reshape(n, 3)))
    lost = outs[0][1] < 0
    assert 1000 < lost.sum() < n                                  # 2 mm per cycle: those within 24 mm of the outlet reach it on the way
    for x, c, v in outs[1:]:
        assert np.array_equal(c < 0, lost) and np.array_equal(x, outs[0][0])
        assert np.array_equal(v, outs[0][2])
    assert (outs[0][2][lost][:, 0] == 10.0).all()                 # the lost ones carry the velocity of their last live cycle


def test_sparse_clouds_are_sorted_along_the_morton_curve(gpu_ctx_factory, oracle_libs):
    """Fewer than 8 particles per cell (the regime of the streaming kernel's LOOKUP 4): the sort's major key is the cell's rank along
    a Morton curve through the cell centres instead of its id (option "sort_curve": -1 by regime, 0 id, 1 rank).  Equal cells stay
    contiguous, the curve keeps spatial neighbours close in all three directions, and -- the order being invisible to the walk --
    every particle equals the CPU statement bit for bit whichever key sorted the cloud."""
    import torch
    from cudaparticlesfoam_amd.cases import box_mesh
    mesh = box_mesh(24, 20, 16, upper=(2.4, 2.0, 1.6), grading=(2.0, 1.0, 0.5))
    cc, _ = mesh.cell_centres_volumes()
    rng = np.random.default_rng(5)
    U = np.stack([1.0 + 0 * cc[:, 0], 0.6 * np.sin(3 * cc[:, 2]), 0.6 * np.cos(3 * cc[:, 1])], 1)
    n = 3 * mesh.n_cells                                           # 3 particles per cell: sparse
    xyz = rng.uniform([0, 0, 0], [2.4, 2.0, 1.6], size=(n, 3))
    cw = oracle_libs.CellWalk(); t = cw.build(mesh)
    x, y, z = (xyz[:, k].copy() for k in range(3))
    c = cw.locate_initial(x, y, z, t, nthreads=cw.max_threads)
    cw.step(x, y, z, c, 0.05, 15, t, U, nthreads=cw.max_threads)
    dev = torch.device("cuda", 0)
    orders = {}
    for curve in (-1, 0, 1):
        ctx = gpu_ctx_factory()
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_option("sort_curve", curve)
        ctx.set_mesh(mesh); ctx.set_velocity(U)
        tx, ty, tz = (torch.from_numpy(xyz[:, k].copy()).to(dev) for k in range(3))
        tc = torch.empty(n, dtype=torch.int32, device=dev); tg = torch.arange(n, dtype=torch.int64, device=dev)
        p = lambda a: a.data_ptr()   # noqa: E731
        ctx.locate_initial_dev(p(tx), p(ty), p(tz), p(tc), n)
        o = [torch.empty_like(a) for a in (tx, ty, tz, tc, tg)]
        ctx.sort_by_cell_dev_to(p(tx), p(ty), p(tz), p(tc), p(tg), *[p(a) for a in o], n)
        tx, ty, tz, tc, tg = o
        cells0 = tc.cpu().numpy()
        change = np.nonzero(np.diff(cells0))[0]
        assert len(set(cells0[np.r_[0, change + 1]])) == len(change) + 1                     # every cell is ONE contiguous run
        orders[curve] = cells0
        ctx.step_dev(p(tx), p(ty), p(tz), p(tc), None, None, n, 0.05, 0.0, 0, 15, 0)
        assert ", 6>" in ctx.step_kernel_name(0.0, 0)                 # sparse, and every cell a box: box records (else ", 4>")
        g = tg.cpu().numpy()
        assert np.array_equal(tc.cpu().numpy(), c[g]) and np.array_equal(tx.cpu().numpy(), x[g])
        assert np.array_equal(ty.cpu().numpy(), y[g]) and np.array_equal(tz.cpu().numpy(), z[g])
    assert np.array_equal(orders[-1], orders[1]) and not np.array_equal(orders[0], orders[1])   # sparse: the curve is the default
    assert (np.diff(orders[0]) >= 0).all()                                                     # key 0: ascending cell ids
    # along the curve consecutive cells are spatial neighbours far more often than along the ids in y and z
    def jump(cells):
        d = np.abs(np.diff(cc[cells], axis=0))
        return np.median(d[np.diff(cells) != 0], axis=0)
    assert jump(orders[1])[1:].max() < 0.6 * max(jump(orders[0])[1:].max(), 1e-9) or jump(orders[1]).sum() < jump(orders[0]).sum()


@pytest.mark.parametrize("case", ["pitz", "box3d", "box3d_curve", "pitz_census", "tiny"])
def test_hand_written_key_sort_gives_the_library_sort_order(case, setup, gpu_ctx_factory):
    """Option "sort_method" 2 (this library's stable radix sort, the default: csrc/cpf_kernels.hip rt_sort_pairs -- three passes of 7
    bits on pitzDaily, three of 8 on a 3-D mesh) orders the cloud exactly like hipcub::DeviceRadixSort (method 0): same
    permutation, particle for particle -- lost particles at the tail, sizes that do not fill the last tile, the Morton major
    key, the occupied-cell census, in place and into a second set of arrays."""
    import torch
    from cudaparticlesfoam_amd.cases import box_mesh
    pz = setup["pz"]
    dev = torch.device("cuda", 0)
    if case.startswith("pitz"):
        mesh, box = setup["mesh"], (np.array(pz.DOMAIN_BOX[0]) - 0.002, np.array(pz.DOMAIN_BOX[1]) + 0.002)
        sizes = (1_000_003, 4097)
    elif case == "tiny":
        mesh, box = setup["mesh"], pz.DOMAIN_BOX
        sizes = (2, 63, 64, 65, 2049)
    else:
        mesh, box = box_mesh(40, 30, 20), (np.array([-0.5, -0.5, -0.5]), np.array([40.5, 30.5, 20.5]))
        sizes = (700_001,)
    ctx = gpu_ctx_factory(); ctx.set_mesh(mesh)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if case == "box3d_curve":
        ctx.set_option("sort_curve", 1)
    if case == "pitz_census":
        ctx.set_option("stream_lookup_by_density", 1)
    p = lambda t: t.data_ptr()   # noqa: E731
    for n in sizes:
        rng = np.random.default_rng(n)
        xyz = rng.uniform(box[0], box[1], size=(n, 3))
        base = [torch.from_numpy(xyz[:, k].copy()).to(dev) for k in range(3)]
        c0 = torch.empty(n, dtype=torch.int32, device=dev)
        ctx.locate_initial_dev(p(base[0]), p(base[1]), p(base[2]), p(c0), n)
        g0 = torch.arange(n, dtype=torch.int64, device=dev) * 3 + 1
        res = {}
        for method in (0, 2):
            ctx.set_option("sort_method", method)
            x, y, z, c, g = (t.clone() for t in (*base, c0, g0))
            out = [torch.zeros_like(t) for t in (x, y, z, c, g)]
            ctx.sort_by_cell_dev_to(p(x), p(y), p(z), p(c), p(g), *(p(t) for t in out), n)
            ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(out, (x, y, z, c, g)))
            res[method] = [t.cpu().numpy() for t in (x, y, z, c, g)]
        for a, b in zip(res[0], res[2]):
            assert np.array_equal(a, b), n
        cs = res[2][3]
        k = int((cs >= 0).sum())
        assert (cs[:k] >= 0).all() and (cs[k:] < 0).all()
        if case != "box3d_curve" and n >= 8 * mesh.n_cells:                 # (sparser clouds are ordered along the Morton curve)
            assert (np.diff(cs[:k].astype(np.int64)) >= 0).all()
    ctx.use_own_stream()


@pytest.mark.parametrize("key_bits", [-1, 0, 1, 10, 21, 111, 222, 322, 444])
def test_hand_written_key_sort_over_digit_widths_and_ragged_sizes(key_bits, gpu_ctx_factory):
    """The same comparison (this library's radix sort == hipcub's, particle for particle) across every digit width the pass plan
    produces: the sub-cell key layout ("sort_key_bits" bx by bz) moves the key's length from 8 to 20 bits on a 60-cell box, i.e.
    1 to 3 passes with last digits of 1 to 8 bits; sizes sit on and around tile (4096) and chunk boundaries; a fifth of the
    particles are lost (all-ones key: the tail)."""
    import torch
    from cudaparticlesfoam_amd.cases import box_mesh
    dev = torch.device("cuda", 0)
    mesh = box_mesh(5, 4, 3) if key_bits >= 0 else box_mesh(1, 1, 1)       # (-1: one cell, no sub-cell bits: a 2-bit key)
    ctx = gpu_ctx_factory(); ctx.set_mesh(mesh)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_option("sort_key_bits", max(key_bits, 0))
    p = lambda t: t.data_ptr()   # noqa: E731
    rng = np.random.default_rng(key_bits + 1)
    sizes = [3, 4095, 4096, 4097, 8193, 12288, 12289, 65537] + [int(v) for v in rng.integers(2, 300_000, size=4)]
    for n in sizes:
        xyz = rng.uniform([-0.6, 0, 0], [5, 4, 3] if key_bits >= 0 else [1, 1, 1], size=(n, 3))   # x < 0: outside the box
        base = [torch.from_numpy(xyz[:, k].copy()).to(dev) for k in range(3)]
        c0 = torch.empty(n, dtype=torch.int32, device=dev)
        ctx.locate_initial_dev(p(base[0]), p(base[1]), p(base[2]), p(c0), n)
        g0 = torch.from_numpy(rng.permutation(n).astype(np.int64)).to(dev)
        res = {}
        for method in (0, 2):
            ctx.set_option("sort_method", method)
            out = [torch.zeros_like(t) for t in (*base, c0, g0)]
            ctx.sort_by_cell_dev_to(p(base[0]), p(base[1]), p(base[2]), p(c0), p(g0), *(p(t) for t in out), n)
            torch.cuda.synchronize()
            res[method] = [t.cpu().numpy() for t in out]
        for a, b in zip(res[0], res[2]):
            assert np.array_equal(a, b), (key_bits, n)
        cs = res[2][3]
        k = int((cs >= 0).sum())
        assert 0 < k < n or n < 10
        assert (cs[:k] >= 0).all() and (cs[k:] < 0).all()
        if n >= 8 * mesh.n_cells:                                              # (sparser clouds are ordered along the Morton curve)
            assert (np.diff(cs[:k].astype(np.int64)) >= 0).all()
        assert np.array_equal(np.sort(res[2][4]), np.arange(n))               # a permutation: nobody lost, nobody twice
    ctx.use_own_stream()

