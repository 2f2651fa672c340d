#!/usr/bin/env python3
"""Developer tool (GPU box): long run of the default step path -- 1e7 particles on pitzDaily, sheared analytic field,
Brownian kick, periodic re-sorts -- then the invariants: nobody lost (every boundary reflects), every particle inside the
cell it claims, and the same final state from the wave-cooperative kernel (variant 3) started from the same cloud.
python tools/soak.py [steps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.cases import pitzdaily as pz
    from cudaparticlesfoam_amd.parallel import x_slab_renumbering
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 10_000_000
    mesh0 = pz.pitzdaily_mesh(); c0, _ = mesh0.cell_centres_volumes()
    mesh = mesh0.renumber_cells(x_slab_renumbering(c0)); centres, _ = mesh.cell_centres_volumes()
    ctx.set_mesh(mesh); ctx.set_velocity(pz.analytic_step_u(mesh, centres))
    x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 4321, dev)
    g = torch.arange(n, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    out = {}
    finals = []
    for variant in (4, 3):
        ctx.set_option("step_variant", variant)
        xs, ys, zs, cs, gs = x.clone(), y.clone(), z.clone(), c.clone(), g.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(0, steps, 100):
            ctx.sort_by_cell_dev(p(xs), p(ys), p(zs), p(cs), p(gs), n)
            D = 1.5e-5 if (s // 100) % 2 else 0.0
            ctx.step_dev(p(xs), p(ys), p(zs), p(cs), p(gs), None, n, 1e-4, D, s, min(100, steps - s), 0)
        torch.cuda.synchronize()
        out["variant_%d_ms_per_step" % variant] = round((time.perf_counter() - t0) / steps * 1e3, 4)
        order = torch.argsort(gs)
        finals.append(tuple(a[order] for a in (xs, ys, zs, cs)))
        if variant == 4:
            out["lost"] = int((cs < 0).sum())
            off, planes, nbr = ctx.mesh_tables()
            idx = torch.randint(0, n, (200000,), device=dev)
            xv, yv, zv, cv = (a[idx].cpu().numpy() for a in (xs, ys, zs, cs))
            pl = planes.reshape(-1, 6, 4)[cv]
            fd = pl[:, :, 3] - (pl[:, :, 0] * xv[:, None] + pl[:, :, 1] * yv[:, None] + pl[:, :, 2] * zv[:, None])
            out["max_plane_distance_outside_own_cell"] = float(fd.max())
    out["variants_4_and_3_identical"] = all(bool(torch.equal(a, b)) for a, b in zip(*finals))
    out["steps"] = steps
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
