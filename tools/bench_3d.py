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
    nbox = tuple(int(k) for k in os.environ.get("CPF_BOX_N", "64,64,60").split(","))     # e.g. 128,128,128: motorBike scale (2.1e6 cells)
    mesh = block_mesh(v, [dict(hex=range(8), n=nbox, simple=(2.0, 1.0, 0.5))])
    tjunction = os.environ.get("CPF_TJUNCTION")           # the reference's TJunction tutorial mesh instead (248 000 cells)
    if tjunction:
        from cudaparticlesfoam_amd.cases import tjunction as tj
        mesh = tj.tjunction_mesh()
    c, _ = mesh.cell_centres_volumes()
    dev = torch.device("cuda", 0)
    ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_mesh(mesh)
    if os.environ.get("CPF_VARIANT"):
        ctx.set_option("step_variant", int(os.environ["CPF_VARIANT"]))
    for kv in os.environ.get("CPF_OPTS", "").split():
        k_, v_ = kv.split("="); ctx.set_option(k_, float(v_))
    if tjunction:
        fields = {"TJunction split flow u0=3": tj.split_flow_u(mesh, c, 0.5), "TJunction split flow u0=5": tj.split_flow_u(mesh, c, 0.5, u0=5.0)}
        box = tj.DOMAIN_BOX
    else:
        box = ((0, 0, 0), (0.3, 0.05, 0.05))
    fields = fields if tjunction else {"diagonal (10,2,1)": np.tile([10.0, 2.0, 1.0], (mesh.n_cells, 1)),
              "swirl": np.stack([10.0 + 0 * c[:, 0], 4 * np.sin(40 * c[:, 2]), 4 * np.cos(40 * c[:, 1])], 1)}
    torch.manual_seed(7)
    if tjunction:
        # uniform over the T: the duct (80 cm^3 ... in mm^3: 80 000) and the cross bar (168 000)
        na = int(n * 80.0 / 248.0)
        u = torch.rand((3, n), dtype=torch.float64, device=dev)
        x = torch.where(torch.arange(n, device=dev) < na, u[0] * 0.2, 0.2 + u[0] * 0.02).contiguous()
        y = torch.where(torch.arange(n, device=dev) < na, -0.01 + u[1] * 0.02, -0.21 + u[1] * 0.42).contiguous()
        z = (u[2] * 0.02).contiguous()
        del u
    else:
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
        fused = int(os.environ.get("CPF_FUSED", "0"))       # 20 cycles in ONE launch (CPF_STEP_FUSE_CYCLES) instead of 20 launches
        ctx.step_dev(p(x), p(y), p(z), p(cell), None, None, n, 1e-4, 0.0, 5, 20, 4 if fused else 0)
        launches, ms = ctx.timing_read(); ctx.timing_enable(False)
        k = ms / launches / (20 if fused else 1)
        print(json.dumps(dict(field=name, cells=mesh.n_cells, particles=n, kernel_ms=round(k, 4),
                              Gparticle_steps_per_s=round(n / k / 1e6, 2), roofline_GBs=round(56 * n / k / 1e6, 1),
                              visits_per_particle_step=round((c1["cells_visited"] - c0["cells_visited"]) /
                                                             max(1, c1["particle_steps"] - c0["particle_steps"]), 3))),
              flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
