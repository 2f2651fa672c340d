"""GPU: BASELINE.json's larger configurations at their full sizes, through size-independent properties plus a
bit-exact comparison with the CPU statement on a subset.

* configs[4] shape (pimple-like, one GPU's share): 440 100-cell mesh (records beyond L2), 1e7 particles, U(t)
  re-uploaded every Eulerian step (src/advect.H:44-57).
* configs[3] size: 1e8 particles (here all on one GPU: what one rank holds at --gpus 1..2 of the scaling run).
* configs[4] "polyMesh": a 3-D mesh that is NOT all-hex at 1e7 particles -- 114 540 cells, a 2:1-refined block inside a
  graded box, 2 242 cells with 9 faces (one face group each) -- with a transient field.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inside_own_cell(ctx, x, y, z, c, sample, dev, torch):
    off, planes, nbr = ctx.mesh_tables()
    idx = torch.randint(0, x.numel(), (sample,), device=dev)
    xs, ys, zs, cs = (t[idx].cpu().numpy() for t in (x, y, z, c))
    pl = planes.reshape(-1, 6, 4)[cs]
    fd = pl[:, :, 3] - (pl[:, :, 0] * xs[:, None] + pl[:, :, 1] * ys[:, None] + pl[:, :, 2] * zs[:, None])
    return float(fd.max())


def test_config5_full_size_transient_velocity(oracle_libs, gpu_ctx_factory, pitz):
    """pitzDaily refined 6 x 6 in-plane = 440 100 hex cells (113 MB of cell records), 1e7 particles, 4 Eulerian
    steps of 6 Lagrangian cycles with a new U before each.  A 1e5-particle subset is carried through the same
    launches and must equal the CPU statement bit for bit; the whole cloud must conserve its count and sit inside
    the cells it claims."""
    import torch
    import bench
    from cudaparticlesfoam_amd.parallel import x_slab_renumbering
    pz = pitz["pz"]
    dev = torch.device("cuda", 0)
    m0 = pz.pitzdaily_mesh(refine=6); c0, _ = m0.cell_centres_volumes()
    mesh = m0.renumber_cells(x_slab_renumbering(c0)); centres, _ = mesh.cell_centres_volumes()
    assert mesh.n_cells == 36 * 12225
    cw = oracle_libs.CellWalk(); t = cw.build(mesh)
    ctx = gpu_ctx_factory()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_mesh(mesh)
    base = pz.analytic_step_u(mesh, centres)
    ctx.set_velocity(base)
    n, ns = 10_000_000, 100_000
    x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 31, dev)
    g = torch.arange(n, dtype=torch.int64, device=dev)
    p = lambda a: a.data_ptr()   # noqa: E731
    ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
    # the subset: every 100th particle of the sorted cloud, followed by its id
    sel = torch.arange(0, n, n // ns, device=dev)[:ns]
    ids = g[sel].cpu().numpy()
    sx, sy, sz, sc = (a[sel].cpu().numpy().copy() for a in (x, y, z, c))
    dt = 1e-4 / 6
    step = 0
    for e in range(4):
        U = base * (1.0 + 0.3 * np.sin(0.7 * e)) + np.array([0.0, 0.4 * np.cos(e), 0.0])
        ctx.set_velocity(U)
        ctx.step_dev(p(x), p(y), p(z), p(c), p(g), None, n, dt, 0.0, step, 6, 0)
        cw.step(sx, sy, sz, sc, dt, 6, t, U, nthreads=cw.max_threads)
        step += 6
        if e == 1:
            ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)       # a re-sort in the middle must not matter
    torch.cuda.synchronize()
    assert int((c >= 0).sum()) == n
    order = torch.argsort(g)
    gx, gy, gz, gc = (a[order][torch.from_numpy(ids).to(dev)].cpu().numpy() for a in (x, y, z, c))
    assert np.array_equal(gc, sc) and np.array_equal(gx, sx) and np.array_equal(gy, sy) and np.array_equal(gz, sz)
    assert _inside_own_cell(ctx, x, y, z, c, 100_000, dev, torch) <= 1e-9
    ctx.use_own_stream()


def test_config4_size_1e8_particles_on_one_gpu(gpu_ctx_factory, pitz):
    """1e8 fp64 particles (3.6 GB of state) on the pitzDaily mesh: count conserved over 6 cycles, every particle of
    a 2e5 sample inside its cell, uniform flow exactly linear where no wall is in reach, and the launch equals
    the same steps done on two halves of the cloud separately (what two ranks would do)."""
    import torch
    import bench
    pz, mesh = pitz["pz"], pitz["mesh"]
    dev = torch.device("cuda", 0)
    ctx = gpu_ctx_factory()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_mesh(mesh); ctx.set_velocity(pitz["U_uniform"])
    n = 100_000_000
    x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 99, dev)
    p = lambda a: a.data_ptr()   # noqa: E731
    ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), None, n)
    h = n // 2 + 37                                                       # an odd split point
    xa, ya, za, ca = x[:h].clone(), y[:h].clone(), z[:h].clone(), c[:h].clone()
    xb, yb, zb, cb = x[h:].clone(), y[h:].clone(), z[h:].clone(), c[h:].clone()
    x0, y0 = x.clone(), y.clone()
    k = 6
    ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, 1e-4, 0.0, 0, k, 0)
    ctx.step_dev(p(xa), p(ya), p(za), p(ca), None, None, h, 1e-4, 0.0, 0, k, 0)
    ctx.step_dev(p(xb), p(yb), p(zb), p(cb), None, None, n - h, 1e-4, 0.0, 0, k, 0)
    torch.cuda.synchronize()
    assert int((c >= 0).sum()) == n
    assert torch.equal(x[:h], xa) and torch.equal(c[:h], ca) and torch.equal(y[h:], yb) and torch.equal(c[h:], cb)
    far = (x0 < 0.2) & (y0.abs() < 0.012) & (x0 > 0.0)
    assert float(((x - x0)[far] - k * 1e-4 * 10.0).abs().max()) < 1e-13 and float((y - y0)[far].abs().max()) == 0.0
    assert _inside_own_cell(ctx, x, y, z, c, 200_000, dev, torch) <= 1e-9
    ctx.use_own_stream()


def test_config5_polyhedral_mesh_1e7_transient_velocity(oracle_libs, gpu_ctx_factory):
    """A graded 40 x 40 x 40 box whose central 20 x 20 x 20 block is refined 2 x 2 x 2: 114 540 cells, the 2 242 unrefined
    cells around the block have 9 faces each -- one face split in four -- (the reference cannot run such a mesh: src/initCuda.H:64).  1e7 particles,
    3 Eulerian steps of 5 cycles with a new swirling U before each, the streaming kernel with mixed records.  A
    1e5-particle subset carried through the same launches must equal the CPU statement bit for bit; the whole cloud must
    conserve its count and sit inside the cells it claims."""
    import torch
    import bench
    from cudaparticlesfoam_amd.cases import refined_box
    dev = torch.device("cuda", 0)
    lo3, hi3 = (0.0, 0.0, 0.0), (0.3, 0.05, 0.05)
    mesh, _ = refined_box(40, 40, 40, lo3, hi3, ((0.075, 0.0125, 0.0125), (0.225, 0.0375, 0.0375)), grading=(2.0, 1.0, 0.5))
    nf = np.diff(mesh.cell_faces()[0])
    assert mesh.n_cells == 114540 and (nf == 9).sum() == 2242 and nf.max() == 9
    cw = oracle_libs.CellWalk(); t = cw.build(mesh)
    assert np.diff(t.cell_off).max() == 6 and t.n_groups > 2000           # six slots per cell, the split faces are groups
    cc, _ = mesh.cell_centres_volumes()
    ctx = gpu_ctx_factory()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_mesh(mesh)
    field = lambda a, b: np.stack([10.0 + 0 * cc[:, 0], a * np.sin(40 * cc[:, 2]), b * np.cos(40 * cc[:, 1])], 1)   # noqa: E731
    ctx.set_velocity(field(4.0, 4.0))
    n, ns = 10_000_000, 100_000
    x, y, z, c = bench.seed_in_fluid(ctx, torch, n, (lo3, hi3), 77, dev)
    g = torch.arange(n, dtype=torch.int64, device=dev)
    p = lambda a: a.data_ptr()   # noqa: E731
    ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
    sel = torch.arange(0, n, n // ns, device=dev)[:ns]
    ids = g[sel].cpu().numpy()
    sx, sy, sz, sc = (a[sel].cpu().numpy().copy() for a in (x, y, z, c))
    assert (nf[sc] > 6).sum() > 500                                        # the subset starts in many-faced cells too
    dt, step = 1e-4, 0
    for e in range(3):
        U = field(4.0 - e, 3.0 + e)
        ctx.set_velocity(U)
        ctx.step_dev(p(x), p(y), p(z), p(c), p(g), None, n, dt, 0.0, step, 5, 0)
        cw.step(sx, sy, sz, sc, dt, 5, t, U, nthreads=cw.max_threads)
        step += 5
    assert ", 11>" in ctx.step_kernel_name(0.0, 0)                 # a refined box: box records with group slots (else ", 3>")
    torch.cuda.synchronize()
    assert int((c >= 0).sum()) == n
    order = torch.argsort(g)
    gx, gy, gz, gc = (a[order][torch.from_numpy(ids).to(dev)].cpu().numpy() for a in (x, y, z, c))
    assert np.array_equal(gc, sc) and np.array_equal(gx, sx) and np.array_equal(gy, sy) and np.array_equal(gz, sz)
    assert _inside_own_cell(ctx, x, y, z, c, 100_000, dev, torch) <= 1e-9
    ctx.use_own_stream()
