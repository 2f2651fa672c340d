#!/usr/bin/env python3
"""Developer tool (GPU box): the step kernel on a mesh that is NOT all-hex -- pitzDaily with the first 60 mm behind the
step refined 2 x 2 x 1 (26 247 cells, 87 of them with 7 faces at the rim of the patch) -- against (a) the same mesh on
the generic CSR walk (what every such mesh ran on before round 3) and (b) the all-hex mesh of the same kind: pitzDaily
refined 2 x 2 x 1 EVERYWHERE (48 900 cells).  1e7 particles, uniform U = (10,0,0), dt 1e-4, sorted cloud.
python tools/bench_mixed.py [particles]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch
    import bench
    from _spinup import device_spinup
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.cases import pitzdaily as pz, refine_hexes, refined_pitzdaily
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    dev = torch.device("cuda", 0)
    m0 = pz.pitzdaily_mesh()
    patch, _ = refined_pitzdaily()
    every, _ = refine_hexes(m0.points, m0.hexes, np.ones(m0.n_cells, bool), split_z=False)
    one = np.zeros(m0.n_cells, bool); one[m0.n_cells // 2] = True
    single, _ = refine_hexes(m0.points, m0.hexes, one, split_z=False)          # what the mixed instantiation itself costs
    cases = [("pitzDaily, patch refined 2x2x1: mixed records (streaming kernel)", patch, {}),
             ("same mesh, generic CSR walk", patch, {"mixed_records": 0}),
             ("pitzDaily refined 2x2x1 everywhere: all-hex", every, {}),
             ("pitzDaily as it is: all-hex", m0, {}),
             ("pitzDaily with ONE cell refined: mixed records (the instantiation's own cost)", single, {})]
    if os.environ.get("CPF_MIXED_3D", "1") != "0":
        # a 3-D mesh: graded 40 x 40 x 40 box with its central 20 x 20 x 20 block refined 2 x 2 x 2 (120 000 cells, the
        # 2 242 unrefined cells around the block have 9 faces), swirling field, particles over the whole box
        from cudaparticlesfoam_amd.cases import box_mesh, refined_box
        lo3, hi3 = (0.0, 0.0, 0.0), (0.3, 0.05, 0.05)
        b3, _ = refined_box(40, 40, 40, lo3, hi3, ((0.075, 0.0125, 0.0125), (0.225, 0.0375, 0.0375)), grading=(2.0, 1.0, 0.5))
        h3 = box_mesh(49, 49, 50, lower=lo3, upper=hi3, grading=(2.0, 1.0, 0.5))          # all-hex, about the same cell count
        one3 = np.zeros(h3.n_cells, bool); one3[h3.n_cells // 2] = True
        s3, _ = refine_hexes(h3.points, h3.hexes, one3, split_z=True)
        cases += [("3-D box, central block refined 2x2x2: mixed records", b3, {}, (lo3, hi3)),
                  ("same mesh, generic CSR walk", b3, {"mixed_records": 0}, (lo3, hi3)),
                  ("3-D box of the same cell count: all-hex", h3, {}, (lo3, hi3)),
                  ("that all-hex box with ONE cell refined: mixed records", s3, {}, (lo3, hi3))]
    if os.environ.get("CPF_MIXED_POLY", "1") != "0":
        # true polyhedra (cases/polygons.py): a 300 x 200 x 4 grid of unit cells in which (a) every ninth square is an octagon
        # (ten planes: two-record cells, 7.7 % of the cells; with triangular prisms and face-grouped neighbours around them: a
        # snappy-like mix), (b) EVERY square has a corner cut (half the cells pentagonal prisms with seven planes: round 3 sent
        # this mesh to the generic walk), (c) every ninth square a dodecagon (fourteen planes: header records), against the
        # all-hex box of about the same cell count
        from cudaparticlesfoam_amd.cases import box_mesh
        from cudaparticlesfoam_amd.cases.polygons import chamfered_box, cut_corner_box
        lo3, hi3 = (0.0, 0.0, 0.0), (300.0, 200.0, 4.0)
        o3, _ = chamfered_box(300, 200, 4, 1)
        p3, _ = cut_corner_box(300, 200, 4, every=1)
        d3, _ = chamfered_box(300, 200, 4, 2)
        hex3 = box_mesh(380, 228, 4, lower=lo3, upper=hi3)
        from cudaparticlesfoam_amd.cases.polygons import diamond_box
        c6, _ = diamond_box(300, 200, 4, 6)
        c1, _ = diamond_box(300, 200, 4, 1)
        hex2 = box_mesh(320, 193, 4, lower=lo3, upper=hi3)                                  # the cell count of the conformal 10 % mesh
        cases += [("CONFORMAL 300x200x4 grid, a diamond at every 6th vertex: 10.5 % of the cells pentagonal prisms (7 planes), no face groups", c6, {}, (lo3, hi3), (3.0, 1.0, 0.3, 0.1)),
                  ("same mesh, generic CSR walk", c6, {"mixed_records": 0}, (lo3, hi3), (3.0, 1.0, 0.3, 0.1)),
                  ("all-hex box of that cell count (320 x 193 x 4)", hex2, {}, (lo3, hi3), (3.0, 1.0, 0.3, 0.1)),
                  ("CONFORMAL truncated-square tiling: half the cells octagonal prisms (10 planes), half diamonds", c1, {}, (lo3, hi3), (3.0, 1.0, 0.3, 0.1)),
                  ("same mesh, generic CSR walk", c1, {"mixed_records": 0}, (lo3, hi3), (3.0, 1.0, 0.3, 0.1))]
        cases += [("300x200x4 grid, every 9th square an octagon (10 planes): two-record cells", o3, {}, (lo3, hi3), (3.0, 1.0, 0.3, 0.1)),
                  ("same mesh, generic CSR walk", o3, {"mixed_records": 0}, (lo3, hi3), (3.0, 1.0, 0.3, 0.1)),
                  ("every square with a cut corner: half the cells pentagonal prisms (7 planes)", p3, {}, (lo3, hi3), (3.0, 1.0, 0.3, 0.1)),
                  ("same mesh, generic CSR walk", p3, {"mixed_records": 0}, (lo3, hi3), (3.0, 1.0, 0.3, 0.1)),
                  ("every 9th square a dodecagon (14 planes): header records", d3, {}, (lo3, hi3), (3.0, 1.0, 0.3, 0.1)),
                  ("all-hex box of about the same cell count (380 x 228 x 4)", hex3, {}, (lo3, hi3), (3.0, 1.0, 0.3, 0.1))]
    only = os.environ.get("CPF_MIXED_ONLY")            # substring filter on the case labels
    for case in cases:
        label, mesh, opts = case[:3]
        if only and only not in label:
            continue
        box3 = case[3] if len(case) > 3 else None
        flow = case[4] if len(case) > 4 else None      # (ux, uy, uz, dt): a uniform drift with a cell-wise wobble
        ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_mesh(mesh)
        if box3 is None:
            ctx.set_velocity(np.tile([10.0, 0.0, 0.0], (mesh.n_cells, 1)))
            x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 1000, dev)
        elif flow is not None:
            cc, _ = mesh.cell_centres_volumes()
            ctx.set_velocity(np.stack([flow[0] + 0 * cc[:, 0], flow[1] * np.sin(0.3 * cc[:, 0]), flow[2] * np.cos(0.2 * cc[:, 1])], 1))
            x, y, z, c = bench.seed_in_fluid(ctx, torch, n, box3, 1000, dev)
        else:
            cc, _ = mesh.cell_centres_volumes()
            ctx.set_velocity(np.stack([10.0 + 0 * cc[:, 0], 4 * np.sin(40 * cc[:, 2]), 4 * np.cos(40 * cc[:, 1])], 1))
            x, y, z, c = bench.seed_in_fluid(ctx, torch, n, box3, 1000, dev)
        dt = flow[3] if flow is not None else 1e-4
        g = torch.arange(n, dtype=torch.int64, device=dev)
        p = lambda t: t.data_ptr()   # noqa: E731
        ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
        ctx.set_option("stats", 1)
        c0 = ctx.counters()
        ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, dt, 0.0, 0, 5, 0)
        torch.cuda.synchronize()
        c1 = ctx.counters()
        ctx.set_option("stats", 0)
        device_spinup(ctx, torch, x, y, z, c, n, dt)
        ctx.timing_enable(True); ctx.timing_read()
        ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, dt, 0.0, 5, 20, 0)
        launches, ms = ctx.timing_read(); ctx.timing_enable(False)
        k = ms / launches
        nf = np.diff(mesh.cell_faces()[0])
        slots = np.diff(ctx.mesh_tables()[0])
        row = dict(case=label, cells=mesh.n_cells, cells_with_more_than_6_faces=int((nf > 6).sum()),
                   cells_with_7_to_12_planes=int(((slots > 6) & (slots <= 12)).sum()), cells_with_more_than_12_planes=int((slots > 12).sum()), particles=n,
                   kernel=ctx.step_kernel_name(0.0, 0), kernel_ms=round(k, 4), Gparticle_steps_per_s=round(n / k / 1e6, 2),
                   roofline_GBs=round(56 * n / k / 1e6, 1),
                   visits_per_particle_step=round((c1["cells_visited"] - c0["cells_visited"]) /
                                                  max(1, c1["particle_steps"] - c0["particle_steps"]), 3))
        print(json.dumps(row), flush=True)
        ctx.close()
        del x, y, z, c, g


if __name__ == "__main__":
    main()
