"""GPU: the Brownian kick (cudaBrownianMotion, cuda/particles.cu:551-599; disp += N(0,1)^3 * sqrt(2 D dt)) over MANY steps.

Parity with the reference's cuRAND XORWOW stream is statistical by contract (no cuRAND here): what the
reference's kick guarantees is a free-space mean square displacement of 6 D t with independent increments and
independent axes, and -- with every boundary reflecting -- that no particle is lost.  Both at the tutorial's
diffusion coefficient (pitzDaily/system/cudaParticlesDict:17-29: diffusionCoeff 1.5e-5, dt 1e-4).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
D, DT = 1.5e-5, 1e-4


@pytest.mark.parametrize("variant", [4, 3])
def test_free_space_msd_is_6Dt(variant, gpu_ctx_factory):
    """200 steps in a box no particle can cross (walls 90 sigma_200 away), zero velocity: <|dr|^2> = 6 D t at
    t = 50, 100, 200 steps (linear growth = increments of different steps are uncorrelated), 2 D t per axis,
    axes uncorrelated, Gaussian kurtosis."""
    from cudaparticlesfoam_amd.cases import box_mesh
    mesh = box_mesh(6, 6, 6, lower=(-0.05, -0.05, -0.05), upper=(0.05, 0.05, 0.05))
    ctx = gpu_ctx_factory()
    ctx.set_option("step_variant", variant)
    ctx.set_mesh(mesh); ctx.set_velocity(np.zeros((mesh.n_cells, 3)))
    n = 400_000
    rng = np.random.default_rng(3)
    start = rng.uniform(-0.02, 0.02, size=(n, 3))              # spread over many cells: the walk crosses faces
    ctx.set_particles(start)
    assert ctx.locate_initial() == 0
    ctx.set_seed(20240607)
    done = 0
    for k in (50, 100, 200):
        ctx.step(DT, D, k - done)
        done = k
        xyzw, cell = ctx.get_particles()
        d = xyzw[:, :3] - start
        assert (cell >= 0).all()
        t = k * DT
        tol = 4.0 * np.sqrt(2.0 / n)                           # 4 sigma of a variance estimate from n samples
        assert abs((d ** 2).sum(1).mean() / (6 * D * t) - 1) < tol, (k, (d ** 2).sum(1).mean() / (6 * D * t))
        assert np.abs(d.var(0) / (2 * D * t) - 1).max() < 1.5 * tol
        assert np.abs(d.mean(0)).max() < 5 * np.sqrt(2 * D * t / n)
    c = np.corrcoef(d.T)
    assert max(abs(c[0, 1]), abs(c[0, 2]), abs(c[1, 2])) < 5 / np.sqrt(n)
    kurt = ((d / d.std(0)) ** 4).mean(0)
    assert np.abs(kurt - 3).max() < 0.05
    assert float(np.abs(d).max()) < 0.03                       # nobody came near a wall: this was free space


def test_count_conserved_with_diffusion_on_pitzdaily(pitz, gpu_ctx_factory):
    """1e6 particles, tutorial D and a 100x larger one, 100 steps of the frozen step flow: every boundary reflects
    (also the front/back planes the kick now pushes particles into), so nobody may be lost, everybody stays in
    the slab, and every particle lies inside the cell it claims."""
    import torch
    import bench
    pz, mesh = pitz["pz"], pitz["mesh"]
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(pitz["U_analytic"])
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda", 0)
    n = 1_000_000
    x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 2718, dev)
    g = torch.arange(n, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    lo, hi = mesh.bounds()
    off, planes, nbr = ctx.mesh_tables()
    step = 0
    for Dk in (D, 100 * D):
        before = ctx.counters()
        ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
        ctx.step_dev(p(x), p(y), p(z), p(c), p(g), None, n, DT, Dk, step, 100, 0)
        step += 100
        torch.cuda.synchronize()
        assert int((c >= 0).sum()) == n and bool(torch.isfinite(x).all() and torch.isfinite(z).all())
        assert float(z.min()) >= lo[2] - 1e-12 and float(z.max()) <= hi[2] + 1e-12
        after = ctx.counters()
        assert after["lost"] == before["lost"] and after["reflections"] > before["reflections"]
        idx = torch.randint(0, n, (50000,), device=dev)
        xs, ys, zs, cs = (t[idx].cpu().numpy() for t in (x, y, z, c))
        pl = planes.reshape(-1, 6, 4)[cs]
        fd = pl[:, :, 3] - (pl[:, :, 0] * xs[:, None] + pl[:, :, 1] * ys[:, None] + pl[:, :, 2] * zs[:, None])
        assert fd.max() <= 1e-9
    assert len(torch.unique(g)) == n                           # ids travelled with the particles through two sorts
    ctx.use_own_stream()


def test_tail_of_the_deviates_reaches_beyond_the_23_bit_cap(oracle_libs, gpu_ctx_factory):
    """The radius uniform of the Box-Muller transform uses all 32 bits of its Philox word (cpf_walk.h normal3):
    among 4e6 particles x 64 steps the CPU statement names the (particle, step) whose radius word is smallest; on the GPU
    that particle's first two deviates must have that radius -- beyond sqrt(2 * 24 * ln 2) = 5.77 sigma, where 23-bit
    uniforms stopped -- and agree with the CPU statement (libm vs the fp32 hardware transcendentals)."""
    import torch
    from cudaparticlesfoam_amd.cases import box_mesh
    cw = oracle_libs.CellWalk()
    n, seed = 4_000_000, 20261003
    gid, step, word = cw.scan_min_radius_word(seed, 0, 64, n)  # 2.6e8 Philox blocks: the smallest word is ~16
    want = cw.normal3(gid, step, seed)
    r_want = float(np.hypot(want[0], want[1]))
    assert word < 2 ** 7 and r_want > 5.9                      # 23-bit uniforms ((w >> 9) + 0.5) gave 5.77 for EVERY word < 512
    mesh = box_mesh(2, 2, 2, lower=(-1, -1, -1), upper=(1, 1, 1))
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(np.zeros((mesh.n_cells, 3)))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_seed(seed)
    dev = torch.device("cuda", 0)
    x = torch.full((n,), 0.5, dtype=torch.float64, device=dev); y = x.clone(); z = x.clone()
    c = torch.empty(n, dtype=torch.int32, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    ctx.locate_initial_dev(p(x), p(y), p(z), p(c), n)
    ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, DT, D, step, 1, 0)      # gid = index; Philox counter (gid, step)
    torch.cuda.synchronize()
    sigma = np.sqrt(2 * D * DT)
    d = np.array([float(x[gid]) - 0.5, float(y[gid]) - 0.5, float(z[gid]) - 0.5]) / sigma
    assert np.abs(d - want).max() < 2e-5, (d, want)
    assert np.hypot(d[0], d[1]) > 5.9
    # and nobody in the whole cloud lies beyond the transform's cap, sqrt(2 * 33 * ln 2) = 6.764 sigma per radius
    r = torch.sqrt((x - 0.5) ** 2 + (y - 0.5) ** 2) / sigma
    assert float(r.max()) < 6.7638 and float(r.max()) >= np.hypot(d[0], d[1]) - 1e-9
    ctx.use_own_stream()


@pytest.mark.parametrize("variant", [4, 3, 0])
def test_front_and_back_planes_mirrored_before_the_walk_equal_the_reference_order(variant, pitz, oracle_libs, gpu_ctx_factory):
    """pitzDaily is one cell thick in z: with the kick the kernels mirror the END POINT about the front / back plane before
    the walk (cpf_walk.h, fold_z) instead of walking to the plane, mirroring there and walking on, as the reference and
    its CPU statement do (ConvexQuery.cu:286-309).  Same trajectory in exact arithmetic, so particle by particle -- the
    ones that bounce off a z plane included -- the two agree to the rounding of the deviates (fp32 hardware transcendentals
    against libm: a few 1e-6 sigma), in position AND cell; with the option off the kernels take the reference's order.
    100x the tutorial's D: a third of the particles meets a z plane in one cycle, some of them twice."""
    pz, mesh = pitz["pz"], pitz["mesh"]
    cw = oracle_libs.CellWalk()
    t = cw.build(mesh)
    U = pitz["U_analytic"]
    n = 200_000
    xyz = pz.uniform_points(5, 260_000, *pz.DOMAIN_BOX)
    c0 = cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t, nthreads=cw.max_threads)
    xyz = xyz[c0 >= 0][:n]
    assert xyz.shape[0] == n
    Db = 100 * D
    sigma = np.sqrt(2 * Db * DT)
    res = {}
    for fold in (1, 0):
        ctx = gpu_ctx_factory()
        ctx.set_option("step_variant", variant); ctx.set_option("z_fold", fold); ctx.set_option("stats", 1)
        ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz)
        assert ctx.locate_initial() == 0
        ctx.set_seed(77)
        before = ctx.counters()
        ctx.step(DT, Db, 1)
        xyzw, cell = ctx.get_particles()
        res[fold] = (xyzw[:, :3].copy(), cell.copy(), ctx.counters()["reflections"] - before["reflections"])
    x, y, z = (xyz[:, k].copy() for k in range(3))
    c = cw.locate_initial(x, y, z, t, nthreads=cw.max_threads)
    stats = cw.step(x, y, z, c, DT, 1, t, U, nthreads=cw.max_threads, D=Db, gid=np.arange(n, dtype=np.int64), step0=0, seed=77)
    ref = np.stack([x, y, z], 1)
    zlo, zhi = mesh.bounds()[0][2], mesh.bounds()[1][2]
    crossed = np.abs(xyz[:, 2] - (zlo + zhi) / 2) + 3 * sigma > (zhi - zlo) / 2          # could have met a z plane
    assert stats[1] > 0.2 * n and crossed.sum() > 0.3 * n
    for fold in (1, 0):
        pos, cell, refl = res[fold]
        assert (cell >= 0).all() and pos[:, 2].min() >= zlo and pos[:, 2].max() <= zhi
        same_cell = cell == c
        assert same_cell.mean() > 0.9999, (fold, float(same_cell.mean()))              # (a 1e-6 sigma shift can move a particle across a face)
        err = np.abs(pos - ref)[same_cell].max()
        assert err < 2e-4 * sigma, (fold, err / sigma)
        assert abs(refl - int(stats[1])) <= 3e-4 * n, (fold, refl, int(stats[1]))      # as many mirrorings as the reference counts
    # and the two orders agree with each other more closely still: same deviates, rounding of one hit point apart
    both = (res[0][1] == res[1][1])
    assert both.mean() > 0.99999 and np.abs(res[0][0] - res[1][0])[both].max() < 1e-12


@pytest.mark.parametrize("variant", [4, 3, 0])
def test_stored_velocity_is_mirrored_with_the_end_point(variant, pitz, gpu_ctx_factory):
    """The velocity an output step stores is mirrored at every wall (ConvexQuery.cu:286-309).  With the end point mirrored
    about the front / back plane before the walk the stored velocity's z flips once per mirroring: same stored velocities
    as with the reference's order, where the particle kept its cell."""
    from cudaparticlesfoam_amd import _lib as L
    pz, mesh = pitz["pz"], pitz["mesh"]
    rng = np.random.default_rng(9)
    U = pitz["U_analytic"].copy(); U[:, 2] = rng.normal(size=U.shape[0]) * 0.5            # a z component to mirror
    xyz = pz.uniform_points(11, 120_000, *pz.DOMAIN_BOX)
    res = {}
    for fold in (1, 0):
        ctx = gpu_ctx_factory()
        ctx.set_option("step_variant", variant); ctx.set_option("z_fold", fold)
        ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz)
        ctx.locate_initial(); ctx.set_seed(5)
        ctx.step(DT, 100 * D, 1, L.STEP_STORE_VEL)
        xyzw, cell, vel = ctx.get_particles(want_vel=True)
        res[fold] = (xyzw[:, :3].copy(), cell.copy(), vel.copy())
    inside = (res[0][1] >= 0) & (res[1][1] == res[0][1])
    assert inside.mean() > 0.9
    flipped = np.sign(res[1][2][inside, 2]) != np.sign(U[np.maximum(res[1][1][inside], 0), 2])
    assert flipped.mean() > 0.05                                       # many stored velocities did get mirrored
    assert np.array_equal(res[1][2][inside], res[0][2][inside])        # and identically in both orders
