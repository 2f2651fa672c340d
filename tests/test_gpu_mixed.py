"""GPU: meshes that are NOT all-hex keep the streaming kernel (BASELINE configs[4] "polyMesh").

The reference cannot run them at all (``src/initCuda.H:64``: ``tetsPerCell = 12``); the bar is this repo's own: the HIP
path is bit-identical to the CPU statement ``oracle/cellwalk.c`` on them -- mesh tables (slots = distinct planes, face
groups for the coplanar pieces of a split face), cells, positions, visit and reflection counters -- whichever kernel runs:
the streaming kernel with mixed cell records (padded records for cells with fewer than six slots, two records for cells with
seven to twelve -- two rounds per visit, both LDS tests --, header records + CSR walk beyond that, face groups resolved at the
exit point), or the generic CSR walk.
"""
import numpy as np
import pytest

from test_oracle_mixed import worst_outside

pytestmark = pytest.mark.gpu


def _prism_box(nx, ny, nz):
    """every hex of a box cut into two triangular prisms (5 faces each): padded records"""
    from cudaparticlesfoam_amd.cases import box_mesh, build_polymesh_from_cells
    m = box_mesh(nx, ny, nz)
    cells = []
    for h in m.hexes:
        for a, b, c in ((0, 1, 2), (0, 2, 3)):
            lo = (h[a], h[b], h[c]); hi = (h[a + 4], h[b + 4], h[c + 4])
            cells.append([(lo[0], lo[2], lo[1]), hi, (lo[0], lo[1], hi[1], hi[0]), (lo[1], lo[2], hi[2], hi[1]),
                          (lo[2], lo[0], hi[0], hi[2])])
    return build_polymesh_from_cells(m.points, [[tuple(int(v) for v in f) for f in c] for c in cells])


def _run_case(oracle_libs, gpu_ctx_factory, mesh, U, xyz, dt, cycles, options, want_kernel, check_inside=True):
    cw = oracle_libs.CellWalk()
    t = cw.build(mesh)
    ref0 = cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t, nthreads=cw.max_threads)
    x, y, z, c = xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), ref0.copy()
    stats = cw.step(x, y, z, c, dt, cycles, t, U, nthreads=cw.max_threads)
    alive = c >= 0                                  # the walk's own invariant: everybody inside the cell they claim
    # (check_inside False: clouds built to sit on vertices and edges, where the reference's walk itself loses the cell)
    assert not check_inside or worst_outside(t, np.stack([x, y, z], 1)[alive], c[alive]).max() <= 1e-9
    for opts in options:
        ctx = gpu_ctx_factory()
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz)
        off, planes, nbr = ctx.mesh_tables()        # the product's own mesh layer against the CPU statement's, bit for bit
        goff, gnbr = ctx.mesh_groups()
        assert np.array_equal(off, t.cell_off) and np.array_equal(nbr, t.nbr) and np.array_equal(planes, t.planes)
        assert np.array_equal(goff, t.group_off) and np.array_equal(gnbr, t.group_nbr[:goff[-1]])
        ctx.locate_initial()
        _, cell0 = ctx.get_particles()
        assert np.array_equal(cell0, ref0), opts
        ctx.sort_by_cell()
        before = ctx.counters()
        ctx.step(dt, 0.0, cycles)
        name = ctx.step_kernel_name(0.0, 0)
        xyzw, cell = ctx.get_particles()
        after = ctx.counters()
        want = want_kernel(opts)
        assert any(w in name for w in ((want,) if isinstance(want, str) else want)), (opts, name)
        assert np.array_equal(cell, c), (opts, int((cell != c).sum()))
        assert np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) and np.array_equal(xyzw[:, 2], z), opts
        assert after["cells_visited"] - before["cells_visited"] == int(stats[0]), opts
        assert after["reflections"] - before["reflections"] == int(stats[1]), opts
    return c


def _kernel_for(opts):
    if opts.get("mixed_records", 1) == 0 or opts.get("step_variant", -1) == 0:
        return "cpf::step_kernel<0,"
    # step_kernel_stream<..., 3>: fixed compare + mixed records, no cell with > 6 slots; 5: the same with the loop lookup
    # (picked above 128 particles per cell, or by the option); 11: a refined mesh whose cells are all axis-aligned boxes walks
    # box records with group slots (tests/test_gpu_box.py) unless a lookup is asked for
    return {0: ", 5>", 1: ", 3>"}.get(opts.get("stream_lookup"), (", 3>", ", 5>", ", 11>"))


def _kernel_for_big(opts):
    if opts.get("mixed_records", 1) == 0 or opts.get("step_variant", -1) == 0:
        return "cpf::step_kernel<0,"
    return ", 2>"                                   # ... 2>: two-record cells (7..12 slots) and header records (more) as well


@pytest.mark.parametrize("seed", [1, 2])
def test_refined_box_cells_with_9_to_21_faces(seed, oracle_libs, gpu_ctx_factory):
    """A graded 3-D box with a 2:1-refined block in the middle: the unrefined cells around it have 9, 12, 15, 18 or 21
    faces -- six slots each, one to five of them face groups of four pieces.  Random cell-constant field, steps that cross
    several cells and bounce off walls, 40 000 particles all over the box (many start in and cross the many-faced
    cells)."""
    from cudaparticlesfoam_amd.cases import refined_box
    mesh, _ = refined_box(8, 6, 5, (0, 0, 0), (8, 6, 5), ((2.0, 1.5, 1.0), (6.0, 4.5, 4.0)), grading=(2.0, 1.0, 0.5))
    off, _ = mesh.cell_faces()
    nf = np.diff(off)
    assert nf.max() >= 18 and (nf == 9).sum() > 20 and (nf == 6).sum() > 300
    rng = np.random.default_rng(seed)
    U = rng.normal(size=(mesh.n_cells, 3)) * 2.0 + np.array([1.0, 0.3, -0.2])
    xyz = rng.uniform([0, 0, 0], [8, 6, 5], size=(40000, 3))
    c = _run_case(oracle_libs, gpu_ctx_factory, mesh, U, xyz, 0.15, 30,
                  [dict(), dict(mixed_records=0), dict(step_variant=0), dict(stream_lookup=0), dict(stream_lookup=1)], _kernel_for)
    assert (c >= 0).all() and (nf[c] > 6).sum() > 500          # every boundary reflects; many END in a many-faced cell


def test_prism_cells_use_padded_records(oracle_libs, gpu_ctx_factory):
    """Cells with FIVE faces (every hex of a box cut into two prisms): padded six-slot records, no CSR walk at all."""
    mesh = _prism_box(7, 6, 5)
    off, _ = mesh.cell_faces()
    assert set(np.diff(off)) == {5}
    rng = np.random.default_rng(11)
    U = rng.normal(size=(mesh.n_cells, 3)) * 1.5
    xyz = rng.uniform([0, 0, 0], [7, 6, 5], size=(30000, 3))
    _run_case(oracle_libs, gpu_ctx_factory, mesh, U, xyz, 0.2, 25, [dict(), dict(step_variant=0)], _kernel_for)


def test_refined_pitzdaily_1e6(oracle_libs, gpu_ctx_factory, pitz):
    """pitzDaily with its first 60 mm behind the step refined 2 x 2 x 1 (26 247 cells, 87 of them with 7 faces), the
    analytic step flow sampled at the new cell centres, 1e6 particles x 20 cycles of the tutorial's dt: bit-identical to
    the CPU statement with the streaming kernel (mixed records) and with the generic walk."""
    from cudaparticlesfoam_amd.cases import refined_pitzdaily
    pz = pitz["pz"]
    mesh, parent = refined_pitzdaily()
    off, _ = mesh.cell_faces()
    nf = np.diff(off)
    assert mesh.n_cells == 26247 and (nf == 7).sum() == 87 and nf.max() == 7
    centres, _ = mesh.cell_centres_volumes()
    U = pz.analytic_step_u(mesh, centres)
    xyz = pz.uniform_points(99, 1_400_000, *pz.DOMAIN_BOX)
    cw = oracle_libs.CellWalk()
    t = cw.build(mesh)
    c0 = cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t, nthreads=cw.max_threads)
    xyz = xyz[c0 >= 0][:1_000_000]
    assert xyz.shape[0] == 1_000_000
    _run_case(oracle_libs, gpu_ctx_factory, mesh, U, xyz, 1e-4, 20, [dict(), dict(stream_lookup=0), dict(step_variant=0)], _kernel_for)


@pytest.mark.parametrize("every", [4, 5])
def test_pentagonal_prisms_are_two_record_cells(every, oracle_libs, gpu_ctx_factory):
    """Cells with SEVEN distinct planes (pentagonal prisms: squares of an extruded grid with a corner cut off) among
    triangular prisms (five: padded records), hexes, and hexes with a hanging node (a face group of two pieces): a lane in a
    seven-slot cell tests the cell's first record in one round and its second (slot 6 + five null planes) in the next."""
    from cudaparticlesfoam_amd.cases.polygons import cut_corner_box
    mesh, kinds = cut_corner_box(11, 8, 3, every=every)
    assert min(kinds.values()) > 0
    rng = np.random.default_rng(40 + every)
    U = rng.normal(size=(mesh.n_cells, 3)) * 1.2
    xyz = rng.uniform([0, 0, 0], [11, 8, 3], size=(40000, 3))
    cw = oracle_libs.CellWalk()
    t = cw.build(mesh)
    slots = np.diff(t.cell_off)
    assert set(slots) == {5, 6, 7} and (slots == 7).sum() * 4 <= mesh.n_cells and t.n_groups > 50
    c = _run_case(oracle_libs, gpu_ctx_factory, mesh, U, xyz, 0.2, 25, [dict(), dict(step_variant=0)], _kernel_for_big)
    assert (c >= 0).all() and (slots[c] == 7).sum() > 1000 and (slots[c] == 5).sum() > 300


def test_mostly_polyhedral_mesh_keeps_the_streaming_kernel(oracle_libs, gpu_ctx_factory):
    """EVERY square of the grid cut: half of the cells are pentagonal prisms with seven (and, with their own hanging nodes,
    face-grouped) slots.  Round 3 sent such a mesh to the generic CSR walk (more than a quarter of the cells had header
    records); with two records per such cell every lane stays on the LDS face test."""
    from cudaparticlesfoam_amd.cases.polygons import cut_corner_box
    mesh, kinds = cut_corner_box(7, 5, 2, every=1)
    assert kinds["pentagons"] == 35 and kinds["triangles"] == 35
    rng = np.random.default_rng(3)
    U = rng.normal(size=(mesh.n_cells, 3)) * 0.8
    xyz = rng.uniform([0, 0, 0], [7, 5, 2], size=(20000, 3))
    _run_case(oracle_libs, gpu_ctx_factory, mesh, U, xyz, 0.2, 20, [dict(), dict(step_variant=0)], _kernel_for_big)


@pytest.mark.parametrize("period", [3, 1])
def test_conformal_polyhedra_without_face_groups(period, oracle_libs, gpu_ctx_factory):
    """cases/polygons.py, diamond_box: pentagonal (period 3) and octagonal (period 1: the truncated square tiling, cells with 7, 8
    and 10 planes at the rim and inside) prisms around diamonds with slanted side faces -- conformal, no hanging nodes: only
    two-record cells and ordinary six-slot ones."""
    from cudaparticlesfoam_amd.cases.polygons import diamond_box
    mesh, kinds = diamond_box(12, 9, 3, period)
    cw = oracle_libs.CellWalk()
    t = cw.build(mesh)
    slots = np.diff(t.cell_off)
    assert t.n_groups == 0 and slots.max() == (7 if period == 3 else 10) and (slots > 6).sum() > 30
    rng = np.random.default_rng(90 + period)
    U = rng.normal(size=(mesh.n_cells, 3)) * 1.2
    xyz = rng.uniform([0, 0, 0], [12, 9, 3], size=(60000, 3))
    c = _run_case(oracle_libs, gpu_ctx_factory, mesh, U, xyz, 0.2, 25, [dict(), dict(step_variant=0)], _kernel_for_big)
    assert (c >= 0).all() and (slots[c] > 6).sum() > 3000


@pytest.mark.parametrize("cuts,want_slots", [(1, 10), (2, 14)])
def test_octagonal_and_dodecagonal_prisms(cuts, want_slots, oracle_libs, gpu_ctx_factory):
    """Squares with all four corners chamfered (cases/polygons.py, chamfered_box): octagonal prisms have TEN distinct planes
    -- two records, four real planes in the second --, dodecagonal ones FOURTEEN: beyond two records, so a header record and
    the walk over the CSR slots.  Around them triangular prisms (padded records) and squares whose shared edge carries two
    hanging nodes (face groups of three pieces).  Unsorted as well: then most lanes of a tile are without a record slot and
    the round walks by per-lane gathers (two-record cells take their CSR slots there)."""
    from cudaparticlesfoam_amd.cases.polygons import chamfered_box
    mesh, kinds = chamfered_box(12, 9, 3, cuts)
    assert kinds["polygons"] == 12 and min(kinds.values()) > 0
    cw = oracle_libs.CellWalk()
    t = cw.build(mesh)
    slots = np.diff(t.cell_off)
    assert set(slots) == {5, 6, want_slots} and (slots == want_slots).sum() == 36
    assert 3 in set(np.diff(t.group_off))
    rng = np.random.default_rng(70 + cuts)
    U = rng.normal(size=(mesh.n_cells, 3)) * 1.2
    xyz = rng.uniform([0, 0, 0], [12, 9, 3], size=(60000, 3))
    c = _run_case(oracle_libs, gpu_ctx_factory, mesh, U, xyz, 0.2, 25, [dict(), dict(step_variant=0)], _kernel_for_big)
    assert (c >= 0).all() and (slots[c] == want_slots).sum() > 3000
    # the same cloud stepped WITHOUT the sort: gather rounds
    x, y, z, cc = xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t)
    cw.step(x, y, z, cc, 0.2, 10, t, U, nthreads=cw.max_threads)
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz)
    ctx.locate_initial()
    ctx.step(0.2, 0.0, 10)
    assert ", 2>" in ctx.step_kernel_name(0.0, 0)
    xyzw, cell = ctx.get_particles()
    assert np.array_equal(cell, cc) and np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) and np.array_equal(xyzw[:, 2], z)


def test_glued_hexes_are_six_slot_cells(oracle_libs, gpu_ctx_factory):
    """Pairs of hexes of a box glued along x: TEN faces each, four coplanar pairs -- six slots, four of them face groups of
    two (towards the glued pairs above, below, in front and behind... which are each ONE cell: both pieces lead to the
    same neighbour, the group holds it twice).  The streaming kernel runs; bit-identical to the CPU statement."""
    from cudaparticlesfoam_amd.cases import box_mesh, build_polymesh_from_cells
    from cudaparticlesfoam_amd.cases.blockmesh import HEX_FACES
    m0 = box_mesh(8, 3, 3)
    cells = []
    for c in range(0, m0.n_cells, 2):                          # cells are numbered i fastest: (c, c + 1) are x neighbours
        a, b = m0.hexes[c], m0.hexes[c + 1]
        loops = [tuple(int(v) for v in a[f]) for k, f in enumerate(HEX_FACES) if k != 1]       # all but a's x+ face
        loops += [tuple(int(v) for v in b[f]) for k, f in enumerate(HEX_FACES) if k != 0]      # all but b's x- face
        cells.append(loops)
    mesh = build_polymesh_from_cells(m0.points, cells)
    assert set(np.diff(mesh.cell_faces()[0])) == {10} and mesh.n_cells == 36
    t = oracle_libs.CellWalk().build(mesh)
    assert set(np.diff(t.cell_off)) == {6}
    rng = np.random.default_rng(3)
    U = rng.normal(size=(mesh.n_cells, 3)) * 0.8
    xyz = rng.uniform([0, 0, 0], [8, 3, 3], size=(20000, 3))
    _run_case(oracle_libs, gpu_ctx_factory, mesh, U, xyz, 0.2, 20, [dict(), dict(step_variant=0)], _kernel_for)


def test_every_lane_of_a_tile_reflects_hit_pool_overflows(oracle_libs, gpu_ctx_factory):
    """The streaming kernel parks wall hit points in a per-wave pool of 10 entries; further ones go to global memory.
    A cloud driven into a corner of a box -- every particle hits a wall every cycle, most of them two or three walls --
    overflows the pool in every tile, every cycle: still bit-identical to the CPU statement, single-cycle and fused
    launches (where the pool is reset per cycle), both record lookups."""
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.cases import box_mesh
    mesh = box_mesh(6, 5, 4)
    cw = oracle_libs.CellWalk()
    t = cw.build(mesh)
    U = np.tile([9.0, 7.0, 5.0], (mesh.n_cells, 1))               # dt * |U| = several cells: to the far corner and back
    rng = np.random.default_rng(17)
    xyz = rng.uniform([0.05, 0.05, 0.05], [5.95, 4.95, 3.95], size=(50000, 3))
    c0 = cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t, nthreads=cw.max_threads)
    x, y, z, c = xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), c0.copy()
    stats = cw.step(x, y, z, c, 0.9, 12, t, U, nthreads=cw.max_threads)
    assert stats[1] > 12 * 50000                                 # more than one reflection per particle-step on average
    for lookup in (0, 1):
        for fused in (0, 1):
            ctx = gpu_ctx_factory()
            ctx.set_option("step_variant", 4); ctx.set_option("stream_lookup", lookup)
            ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz)
            ctx.locate_initial(); ctx.sort_by_cell()
            before = ctx.counters()
            ctx.step(0.9, 0.0, 12, L.STEP_FUSE_CYCLES if fused else 0)
            xyzw, cell = ctx.get_particles()
            after = ctx.counters()
            assert "step_kernel_stream" in ctx.step_kernel_name(0.0, 0)
            assert np.array_equal(cell, c), (lookup, fused)
            assert np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) and np.array_equal(xyzw[:, 2], z)
            assert after["reflections"] - before["reflections"] == int(stats[1])
            assert after["lost"] - before["lost"] == int(stats[2])


@pytest.mark.parametrize("which", ["refined_box", "cut_corners", "octagons"])
def test_diffusion_on_a_mixed_mesh_loses_nobody(which, oracle_libs, gpu_ctx_factory):
    """The Brownian kick on the refined box (LOOKUP = 3 with the kick: face groups, hit points in the per-wave pool) and
    on the cut-corner grid and the chamfered grid (LOOKUP = 2: two-record cells as well).  Parity with the CPU statement is statistical there, so
    the check is the domain's own -- every boundary reflects, so after 60 kicked cycles nobody is lost and every particle
    lies inside the cell it claims (all plane distances <= 0), many-faced cells included."""
    if which == "refined_box":
        from cudaparticlesfoam_amd.cases import refined_box
        mesh, _ = refined_box(8, 6, 5, (0, 0, 0), (8, 6, 5), ((2.0, 1.5, 1.0), (6.0, 4.5, 4.0)), grading=(2.0, 1.0, 0.5))
        hi, want = [8, 6, 5], ", 11>"                                 # every cell a box: box records with group slots (round 3: ", 5>")
    elif which == "cut_corners":
        from cudaparticlesfoam_amd.cases.polygons import cut_corner_box
        mesh, _ = cut_corner_box(11, 8, 3, every=4)
        hi, want = [11, 8, 3], ", 2>"
    else:
        from cudaparticlesfoam_amd.cases.polygons import chamfered_box
        mesh, _ = chamfered_box(12, 9, 3, 1)                          # ten-slot cells: both records hold real planes
        hi, want = [12, 9, 3], ", 2>"
    cw = oracle_libs.CellWalk()
    t = cw.build(mesh)
    rng = np.random.default_rng(23)
    n = 200_000
    xyz = rng.uniform([0, 0, 0], hi, size=(n, 3))
    U = rng.normal(size=(mesh.n_cells, 3)) * 0.5
    ctx = gpu_ctx_factory()
    ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz)
    assert ctx.locate_initial() == 0
    ctx.sort_by_cell()
    before = ctx.counters()
    ctx.step(0.05, 0.4, 60)                                        # sigma = sqrt(2 D dt) = 0.2 cell widths per cycle
    assert want in ctx.step_kernel_name(0.4, 0)
    xyzw, cell = ctx.get_particles()
    after = ctx.counters()
    assert (cell >= 0).all() and after["lost"] == before["lost"] and after["reflections"] > before["reflections"]
    nf = np.diff(mesh.cell_faces()[0])
    assert (nf[cell] > 6).sum() > 1000                             # plenty of them ended in a many-faced cell
    w = worst_outside(t, xyzw[:, :3], cell)
    assert w.max() <= 1e-9, (int((w > 1e-9).sum()), float(w.max()))
    assert float(np.abs(xyzw[:, :3] - xyz).max()) > 0.5            # they did move
