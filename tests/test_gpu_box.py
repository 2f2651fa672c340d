"""GPU: meshes whose cells are all axis-aligned boxes walk BOX RECORDS (csrc/cpf_walk.h "box records"; the streaming kernel's
LOOKUP 6 instantiation) -- 128-byte records without normals, three candidate faces per visit instead of six.

The bar is the usual one: bit-identical to the CPU statement ``oracle/cellwalk.c`` (which knows nothing of boxes: it runs the
reference's predicate on every face, ``third_party/RTXAdvect/query/ConvexQuery.cu:32-131``) -- cells, positions, visit and
reflection counters -- and bit-identical to the same library with ``box_records`` 0.  The structured clouds are the point: the
three-candidate form hands a wave to the six-face form whenever a lane sits outside a face it did not come in through, or two
axes tie in dT (the walk's ORIGINAL slot order decides), and particles on faces, edges, vertices and cell diagonals of a uniform
grid do exactly that.
"""
import numpy as np
import pytest

from test_gpu_mixed import _run_case

pytestmark = pytest.mark.gpu


def _kernel(opts):
    if opts.get("step_variant", -1) == 0:
        return "cpf::step_kernel<0,"
    if opts.get("box_records", 1) == 0:
        # (", 9>": a field without a z component on a box mesh is a flat case too -- csrc/cpf_walk.h "flat walk")
        return {0: ", 0>", 4: ", 4>"}.get(opts.get("stream_lookup"), (", 1>", ", 9>"))
    return {0: ", 0>", 1: ", 1>", 4: ", 4>"}.get(opts.get("stream_lookup"), ", 6>")


OPTS = [{}, {"box_records": 0}, {"stream_lookup": 6}, {"stream_lookup": 4}, {"stream_lookup": 1}]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_box_records_random_cloud(oracle_libs, gpu_ctx_factory, seed):
    from cudaparticlesfoam_amd.api import mesh_box_records_host
    from cudaparticlesfoam_amd.cases import box_mesh
    rng = np.random.default_rng(seed)
    mesh = box_mesh(12, 10, 8, lower=(-0.3, 0.1, 0.0), upper=(0.9, 0.85, 0.64), grading=(3.0, 0.5, 1.0))
    assert mesh_box_records_host(mesh) is not None
    cc, _ = mesh.cell_centres_volumes()
    U = rng.normal(size=(mesh.n_cells, 3)) * np.array([1.5, 1.0, 0.8])
    if seed == 2:
        U[:, 2] = 0.0                                          # nothing moves in z: the axis drops out wave-uniformly
    if seed == 3:
        U[rng.random(mesh.n_cells) < 0.3] = 0.0                # cells at rest
    n = 20000
    xyz = np.array([-0.3, 0.1, 0.0]) + rng.random((n, 3)) * np.array([1.2, 0.75, 0.64])
    _run_case(oracle_libs, gpu_ctx_factory, mesh, U, xyz, 0.11, 6, OPTS, _kernel)


def _structured_cloud(nx, ny, nz, rng):
    """cell centres, face centres, edge mid points, vertices and points an ulp or two off them, of a unit grid"""
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    pts = [g + 0.5]
    for off in ((0.0, 0.5, 0.5), (0.5, 0.0, 0.5), (0.5, 0.5, 0.0), (0.0, 0.0, 0.5), (0.0, 0.5, 0.0), (0.5, 0.0, 0.0), (0.0, 0.0, 0.0),
                (0.25, 0.25, 0.25), (0.75, 0.25, 0.5)):
        pts.append(g + np.array(off))
    p = np.concatenate(pts)
    p = p[(p > 0).all(1) & (p[:, 0] < nx) & (p[:, 1] < ny) & (p[:, 2] < nz)]     # (strictly inside the domain)
    nudged = p.copy()
    k = rng.integers(-2, 3, size=p.shape)
    for _ in range(2):
        nudged = np.where(k > 0, np.nextafter(nudged, np.inf), np.where(k < 0, np.nextafter(nudged, -np.inf), nudged))
        k = k - np.sign(k)
    return np.concatenate([p, nudged])


@pytest.mark.parametrize("field", ["diag", "diag_neg", "xy", "x", "signs", "half"])
def test_box_records_ties_and_faces(oracle_libs, gpu_ctx_factory, field):
    from cudaparticlesfoam_amd.cases import box_mesh
    rng = np.random.default_rng(11)
    nx, ny, nz = 8, 7, 6
    mesh = box_mesh(nx, ny, nz)
    cc, _ = mesh.cell_centres_volumes()
    one = np.ones(mesh.n_cells)
    U = {"diag": np.stack([one, one, one], 1), "diag_neg": -np.stack([one, one, one], 1), "xy": np.stack([one, -one, 0 * one], 1),
         "x": np.stack([one, 0 * one, 0 * one], 1),
         "signs": np.sign(np.sin(1.7 * cc + np.array([0.3, 1.1, 2.0]))),        # +-1 per axis, changing from cell to cell
         "half": 0.5 * np.sign(np.cos(2.3 * cc[:, [1, 2, 0]]))}[field]
    xyz = _structured_cloud(nx, ny, nz, rng)
    # dt = 1: a centre goes to the next centre through the shared vertex (three equal dT), a face point to the next face ...
    _run_case(oracle_libs, gpu_ctx_factory, mesh, U, xyz, 1.0, 4, [{"stream_lookup": 6}, {"box_records": 0, "stream_lookup": 1}], _kernel,
              check_inside=False)


def test_box_records_signed_zero_normals(oracle_libs, gpu_ctx_factory):
    """Coordinates and velocity components that are exactly -0.0 keep their sign through a wall reflection exactly as with the
    full planes (the box record remembers the sign of every zero component of every normal)."""
    from cudaparticlesfoam_amd.cases import box_mesh
    rng = np.random.default_rng(5)
    mesh = box_mesh(6, 5, 4, lower=(-3.0, -2.0, -2.0))           # the planes x = 0, y = 0, z = 0 are cell faces
    U = rng.normal(size=(mesh.n_cells, 3)) * 2.0
    U[::3, 2] = -0.0
    U[1::3, 1] = -0.0
    n = 6000
    xyz = np.array([-3.0, -2.0, -2.0]) + rng.random((n, 3)) * np.array([6.0, 5.0, 4.0])
    xyz[::5, 2] = -0.0
    xyz[1::5, 1] = -0.0
    cw = oracle_libs.CellWalk()
    t = cw.build(mesh)
    ref0 = cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t, nthreads=cw.max_threads)
    x, y, z, c = xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), ref0.copy()
    cw.step(x, y, z, c, 0.4, 5, t, U, nthreads=cw.max_threads)
    outs = []
    for opts in ({"stream_lookup": 6}, {"box_records": 0, "stream_lookup": 1}):
        ctx = gpu_ctx_factory()
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz); ctx.locate_initial(); ctx.sort_by_cell()
        ctx.step(0.4, 0.0, 5, 2)                                   # CPF_STEP_STORE_VEL
        xyzw, cell, vel = ctx.get_particles(want_vel=True)
        outs.append((xyzw.copy(), cell.copy(), vel.copy()))
        assert np.array_equal(cell, c)
        # BITS, not values: -0.0 == 0.0 would hide exactly what this test is about
        assert np.array_equal(xyzw[:, :3].view(np.int64), np.stack([x, y, z], 1).view(np.int64)), opts
    assert np.array_equal(outs[0][2].view(np.int64), outs[1][2].view(np.int64))
    assert np.array_equal(outs[0][0].view(np.int64), outs[1][0].view(np.int64))


@pytest.mark.parametrize("D", [1e-3])
def test_box_records_brownian_same_bits_as_full_records(gpu_ctx_factory, D):
    """With the kick there is no CPU statement to be bit-identical to (the generator differs: SURVEY.md 8c) -- but box records
    and full records run the same generator and must agree bit for bit."""
    from cudaparticlesfoam_amd.cases import box_mesh
    rng = np.random.default_rng(9)
    mesh = box_mesh(10, 9, 8, upper=(1.0, 0.9, 0.8), grading=(2.0, 1.0, 0.5))
    U = rng.normal(size=(mesh.n_cells, 3))
    xyz = rng.random((30000, 3)) * np.array([1.0, 0.9, 0.8])
    outs = []
    for opts in ({}, {"box_records": 0}):
        ctx = gpu_ctx_factory()
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz); ctx.locate_initial(); ctx.sort_by_cell()
        ctx.set_seed(77)
        ctx.step(0.05, D, 6)
        assert (", 6>" if not opts else ", 1>") in ctx.step_kernel_name(D, 0)
        xyzw, cell = ctx.get_particles()
        outs.append((xyzw.copy(), cell.copy(), ctx.counters()))
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][0].view(np.int64), outs[1][0].view(np.int64))
    assert outs[0][2] == outs[1][2]


def _refined_unit_box(nx, ny, nz, region):
    from cudaparticlesfoam_amd.cases import refined_box
    return refined_box(nx, ny, nz, (0.0, 0.0, 0.0), (float(nx), float(ny), float(nz)), region)[0]


@pytest.mark.parametrize("seed", [1, 2])
def test_refined_box_random_cloud(oracle_libs, gpu_ctx_factory, seed):
    """A 2:1-refined box (face groups around the refined block: castellated snappyHexMesh output).  Its cells are all axis-aligned
    boxes too -- a coarse box next to the block keeps ONE slot for its split face, neighbour code = the group's -- so the mesh
    gets box records and the LOOKUP 11 instantiation: box records + group slots (a group is only ever left through the face the
    particle moves towards).  Bit-identical to the CPU statement and to the ordinary mixed records, sparse and dense."""
    from cudaparticlesfoam_amd.cases import refined_box
    rng = np.random.default_rng(seed)
    mesh, _ = refined_box(10, 9, 8, (-0.2, 0.0, 0.1), (1.0, 0.9, 0.9), ((0.1, 0.2, 0.3), (0.7, 0.7, 0.7)), grading=(2.0, 1.0, 0.5))
    U = rng.normal(size=(mesh.n_cells, 3)) * np.array([1.5, 1.0, 0.8])
    if seed == 2:
        U[:, 1] = 0.0
    n = 30000 if seed == 1 else 3000                            # dense / sparse (fewer than 8 per cell)
    xyz = np.array([-0.2, 0.0, 0.1]) + rng.random((n, 3)) * np.array([1.2, 0.9, 0.8])
    opts = [{}, {"box_records": 0}, {"stream_lookup": 0}, {"box_records": 0, "stream_lookup": 0}]
    _run_case(oracle_libs, gpu_ctx_factory, mesh, U, xyz, 0.09, 6, opts,
              lambda o: ", 5>" if o.get("stream_lookup") == 0 else (", 3>" if o.get("box_records", 1) == 0 else ", 11>"))


@pytest.mark.parametrize("field", ["diag", "signs", "half"])
def test_refined_box_ties_and_faces(oracle_libs, gpu_ctx_factory, field):
    """... and the structured clouds of test_box_records_ties_and_faces on a unit grid with a refined block: particles on the
    faces, edges and vertices of coarse AND fine cells, on the pieces of split faces, equal dT on several axes."""
    rng = np.random.default_rng(13)
    nx, ny, nz = 8, 6, 6
    mesh = _refined_unit_box(nx, ny, nz, ((2.0, 1.0, 1.0), (6.0, 5.0, 5.0)))
    cc, _ = mesh.cell_centres_volumes()
    one = np.ones(mesh.n_cells)
    U = {"diag": np.stack([one, one, one], 1), "signs": np.sign(np.sin(1.7 * cc + np.array([0.3, 1.1, 2.0]))),
         "half": 0.5 * np.sign(np.cos(2.3 * cc[:, [1, 2, 0]]))}[field]
    coarse = _structured_cloud(nx, ny, nz, rng)
    fine = 0.5 * _structured_cloud(2 * nx, 2 * ny, 2 * nz, rng)             # the fine cells' centres, faces, edges, vertices
    xyz = np.concatenate([coarse, fine[rng.random(len(fine)) < 0.3]])
    _run_case(oracle_libs, gpu_ctx_factory, mesh, U, xyz, 1.0, 4, [{}, {"box_records": 0}],
              lambda o: ", 3>" if o.get("box_records", 1) == 0 else ", 11>", check_inside=False)


def test_refined_box_kick_and_velocity_refresh_same_bits(gpu_ctx_factory):
    """(with the kick there is no CPU statement to match: box records with group slots against the ordinary mixed records)"""
    from cudaparticlesfoam_amd.cases import refined_box
    rng = np.random.default_rng(3)
    mesh, _ = refined_box(10, 9, 8, (0.0, 0.0, 0.0), (1.0, 0.9, 0.8), ((0.2, 0.2, 0.2), (0.7, 0.7, 0.6)))
    U = rng.normal(size=(mesh.n_cells, 3))
    xyz = rng.random((40000, 3)) * np.array([1.0, 0.9, 0.8])
    outs = []
    for opts in ({}, {"box_records": 0}):
        ctx = gpu_ctx_factory()
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz); ctx.locate_initial(); ctx.sort_by_cell()
        ctx.set_seed(5)
        ctx.step(0.04, 1e-3, 6)
        U2 = U[::-1].copy()
        ctx.set_velocity(U2)                                    # (a velocity refresh between two launches)
        ctx.step(0.04, 1e-3, 3, 2)
        xyzw, cell, vel = ctx.get_particles(want_vel=True)
        outs.append((xyzw.copy(), cell.copy(), vel.copy(), ctx.counters()))
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][0].view(np.int64), outs[1][0].view(np.int64))
    assert np.array_equal(outs[0][2].view(np.int64), outs[1][2].view(np.int64))
    assert outs[0][3] == outs[1][3]
