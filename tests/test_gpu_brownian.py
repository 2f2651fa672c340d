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
