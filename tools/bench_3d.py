#!/usr/bin/env python3
"""Developer tool (GPU box): the step kernel on a genuinely 3-D mesh (TJunction scale: 64 x 64 x 60 = 245 760 graded
hex cells, records 63 MB: beyond L2), diagonal and swirling cell-constant fields, 1e7 particles."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.cases import block_mesh
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    v = np.array([[0, 0, 0], [0.3, 0, 0], [0.3, 0.05, 0], [0, 0.05, 0], [0, 0, 0.05], [0.3, 0, 0.05], [0.3, 0.05, 0.05],
                  [0, 0.05, 0.05]], float)
    mesh = block_mesh(v, [dict(hex=range(8), n=(64, 64, 60), simple=(2.0, 1.0, 0.5))])
    c, _ = mesh.cell_centres_volumes()
    dev = torch.device("cuda", 0)
    ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_mesh(mesh)
    if os.environ.get("CPF_VARIANT"):
        ctx.set_option("step_variant", int(os.environ["CPF_VARIANT"]))
    for kv in os.environ.get("CPF_OPTS", "").split():
        k_, v_ = kv.split("="); ctx.set_option(k_, float(v_))
    fields = {"diagonal (10,2,1)": np.tile([10.0, 2.0, 1.0], (mesh.n_cells, 1)),
              "swirl": np.stack([10.0 + 0 * c[:, 0], 4 * np.sin(40 * c[:, 2]), 4 * np.cos(40 * c[:, 1])], 1)}
    torch.manual_seed(7)
    x = torch.rand(n, dtype=torch.float64, device=dev) * 0.3
    y = torch.rand(n, dtype=torch.float64, device=dev) * 0.05
    z = torch.rand(n, dtype=torch.float64, device=dev) * 0.05
    cell = torch.empty(n, dtype=torch.int32, device=dev); gid = torch.arange(n, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    for name, U in fields.items():
        ctx.set_velocity(U)
        ctx.locate_initial_dev(p(x), p(y), p(z), p(cell), n)
        ctx.sort_by_cell_dev(p(x), p(y), p(z), p(cell), p(gid), n)
        ctx.set_option("stats", 1)
        c0 = ctx.counters()
        ctx.step_dev(p(x), p(y), p(z), p(cell), None, None, n, 1e-4, 0.0, 0, 5, 0)
        torch.cuda.synchronize()
        c1 = ctx.counters()
        ctx.set_option("stats", 0)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from _spinup import device_spinup
        device_spinup(ctx, torch, x, y, z, cell, n, 1e-4)
        ctx.timing_enable(True); ctx.timing_read()
        ctx.step_dev(p(x), p(y), p(z), p(cell), None, None, n, 1e-4, 0.0, 5, 20, 0)
        launches, ms = ctx.timing_read(); ctx.timing_enable(False)
        k = ms / launches
        print(json.dumps(dict(field=name, cells=mesh.n_cells, particles=n, kernel_ms=round(k, 4),
                              Gparticle_steps_per_s=round(n / k / 1e6, 2), roofline_GBs=round(56 * n / k / 1e6, 1),
                              visits_per_particle_step=round((c1["cells_visited"] - c0["cells_visited"]) /
                                                             max(1, c1["particle_steps"] - c0["particle_steps"]), 3))),
              flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
