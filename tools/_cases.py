"""Developer tools (GPU box): the synthetic cases the tools share -- one place for mesh, field and cloud set-up.
  pitz       pitzDaily 12 225 cells (x-slab numbering), uniform (10,0,0) or the analytic step flow
  box3d      graded 64 x 64 x 60 box = 245 760 hex cells (records 63 MB: beyond L2), diagonal / swirl fields; CPF_BOX_N=a,b,c resizes
  refbox3d   the same box (default 40 x 40 x 40) with its central block refined 2 x 2 x 2: face groups, mixed records
  tjunction  the reference's TJunction tutorial mesh, 248 000 cells of 1 mm, closed-form split flow (u0 = 3 / 5), cloud over the whole T
  tjunction_run   ... seeded as the tutorial's dictionary does: in the first 50 mm of the inlet duct
  octagons / pentagons / dodecagons / hexgrid   300 x 200 x 4 unit cells (cases/polygons.py): every ninth square an octagonal prism
             (10 planes: two-record cells) / every square with a cut corner (half the cells 7 planes) / every ninth a dodecagonal
             prism (14 planes: header records) / the all-hex box of about the same cell count; dt = 0.1 (POLY_DT)
  diamonds6 / diamonds1 / hexgrid6   CONFORMAL polyhedra (no hanging nodes): a diamond at every 6th grid vertex (10.5 % of the cells
             pentagonal prisms, 7 planes) / at every vertex (truncated square tiling: half the cells octagonal prisms, 10 planes)
             / the all-hex box with diamonds6's cell count"""
import os

import numpy as np

BOX = ((0.0, 0.0, 0.0), (0.3, 0.05, 0.05))
POLY_DT = 0.1
POLY_CASES = ("octagons", "pentagons", "dodecagons", "hexgrid", "diamonds6", "diamonds1", "hexgrid6")


def make_case(name, ctx, torch, n, dev, field=None, seed=7):
    """Sets mesh and field on ctx; returns (mesh, x, y, z, cell, fields) with the cloud located (not sorted);
    fields = {label: U} of the case (the first one, or `field`, is the one set)."""
    if name == "pitz":
        import bench
        from cudaparticlesfoam_amd.cases import pitzdaily as pz
        from cudaparticlesfoam_amd.parallel import x_slab_renumbering
        mesh0 = pz.pitzdaily_mesh(); c0, _ = mesh0.cell_centres_volumes()
        mesh = mesh0.renumber_cells(x_slab_renumbering(c0)); cc, _ = mesh.cell_centres_volumes()
        fields = {"uniform": pz.uniform_u(mesh), "analytic": pz.analytic_step_u(mesh, cc)}
        ctx.set_mesh(mesh)
        ctx.set_velocity(fields[field or "uniform"])
        x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 1000, dev)
        return mesh, x, y, z, c, fields
    if name in POLY_CASES:
        import bench
        from cudaparticlesfoam_amd.cases import box_mesh
        from cudaparticlesfoam_amd.cases.polygons import chamfered_box, cut_corner_box, diamond_box
        lo3, hi3 = (0.0, 0.0, 0.0), (300.0, 200.0, 4.0)
        mesh = {"octagons": lambda: chamfered_box(300, 200, 4, 1)[0], "pentagons": lambda: cut_corner_box(300, 200, 4, every=1)[0],
                "dodecagons": lambda: chamfered_box(300, 200, 4, 2)[0],
                "diamonds6": lambda: diamond_box(300, 200, 4, 6)[0], "diamonds1": lambda: diamond_box(300, 200, 4, 1)[0],
                "hexgrid6": lambda: box_mesh(320, 193, 4, lower=lo3, upper=hi3),
                "hexgrid": lambda: box_mesh(380, 228, 4, lower=lo3, upper=hi3)}[name]()
        cc, _ = mesh.cell_centres_volumes()
        fields = {"drift": np.stack([3.0 + 0 * cc[:, 0], 1.0 * np.sin(0.3 * cc[:, 0]), 0.3 * np.cos(0.2 * cc[:, 1])], 1)}
        ctx.set_mesh(mesh)
        ctx.set_velocity(fields["drift"])
        x, y, z, c = bench.seed_in_fluid(ctx, torch, n, (lo3, hi3), 1000, dev)
        return mesh, x, y, z, c, fields
    torch.manual_seed(seed)
    if name == "tjunction_run":
        # the tutorial as its dictionary seeds it: all particles in the first 50 mm of the inlet duct (200 per cell at 4e6)
        import bench
        from cudaparticlesfoam_amd.cases import tjunction as tj
        mesh = tj.tjunction_mesh(); cc, _ = mesh.cell_centres_volumes()
        fields = {"u0=3": tj.split_flow_u(mesh, cc, 0.5)}
        ctx.set_mesh(mesh); ctx.set_velocity(fields["u0=3"])
        x, y, z, c = bench.seed_in_fluid(ctx, torch, n, tj.PARTICLE_DICT["seedingBox"], 2027, dev)
        return mesh, x, y, z, c, fields
    if name == "tjunction":
        from cudaparticlesfoam_amd.cases import tjunction as tj
        mesh = tj.tjunction_mesh(); cc, _ = mesh.cell_centres_volumes()
        fields = {"u0=3": tj.split_flow_u(mesh, cc, 0.5), "u0=5": tj.split_flow_u(mesh, cc, 0.5, u0=5.0)}
        na = int(n * 80.0 / 248.0)                          # uniform over the T: the duct (80 cm^3) and the cross bar (168)
        u = torch.rand((3, n), dtype=torch.float64, device=dev)
        ar = torch.arange(n, device=dev)
        x = torch.where(ar < na, u[0] * 0.2, 0.2 + u[0] * 0.02).contiguous()
        y = torch.where(ar < na, -0.01 + u[1] * 0.02, -0.21 + u[1] * 0.42).contiguous()
        z = (u[2] * 0.02).contiguous()
        del u, ar
    elif name == "box3d":
        from cudaparticlesfoam_amd.cases import block_mesh
        v = np.array([[0, 0, 0], [0.3, 0, 0], [0.3, 0.05, 0], [0, 0.05, 0], [0, 0, 0.05], [0.3, 0, 0.05], [0.3, 0.05, 0.05],
                      [0, 0.05, 0.05]], float)
        nbox = tuple(int(k) for k in os.environ.get("CPF_BOX_N", "64,64,60").split(","))
        mesh = block_mesh(v, [dict(hex=range(8), n=nbox, simple=(2.0, 1.0, 0.5))]); cc, _ = mesh.cell_centres_volumes()
        fields = {"diagonal": np.tile([10.0, 2.0, 1.0], (mesh.n_cells, 1)),
                  "swirl": np.stack([10.0 + 0 * cc[:, 0], 4 * np.sin(40 * cc[:, 2]), 4 * np.cos(40 * cc[:, 1])], 1)}
        x = torch.rand(n, dtype=torch.float64, device=dev) * 0.3
        y = torch.rand(n, dtype=torch.float64, device=dev) * 0.05
        z = torch.rand(n, dtype=torch.float64, device=dev) * 0.05
    elif name == "refbox3d":
        # the 3-D box with its central block (half the extent per axis) refined 2 x 2 x 2: face groups around the block, the
        # snappyHexMesh kind of mesh; CPF_BOX_N as for box3d (default 40,40,40 -> 120 000 cells)
        from cudaparticlesfoam_amd.cases import refined_box
        nbox = tuple(int(k) for k in os.environ.get("CPF_BOX_N", "40,40,40").split(","))
        mesh, _ = refined_box(*nbox, (0.0, 0.0, 0.0), (0.3, 0.05, 0.05), ((0.075, 0.0125, 0.0125), (0.225, 0.0375, 0.0375)),
                              grading=(2.0, 1.0, 0.5))
        cc, _ = mesh.cell_centres_volumes()
        fields = {"diagonal": np.tile([10.0, 2.0, 1.0], (mesh.n_cells, 1)),
                  "swirl": np.stack([10.0 + 0 * cc[:, 0], 4 * np.sin(40 * cc[:, 2]), 4 * np.cos(40 * cc[:, 1])], 1)}
        x = torch.rand(n, dtype=torch.float64, device=dev) * 0.3
        y = torch.rand(n, dtype=torch.float64, device=dev) * 0.05
        z = torch.rand(n, dtype=torch.float64, device=dev) * 0.05
    else:
        raise ValueError(name)
    ctx.set_mesh(mesh)
    ctx.set_velocity(fields[field or next(iter(fields))])
    c = torch.empty(n, dtype=torch.int32, device=dev)
    ctx.locate_initial_dev(x.data_ptr(), y.data_ptr(), z.data_ptr(), c.data_ptr(), n)
    return mesh, x, y, z, c, fields
