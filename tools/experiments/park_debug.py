#!/usr/bin/env python3
"""Developer tool (GPU box, library built with -DCPF_PARK_DEBUG): straggler-parking census on the bench cloud.
With that build the statistics counters mean: particle_steps = particles + rounds run by regular tiles, reflections =
rounds run by batches, lost = parked particles."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from cudaparticlesfoam_amd.api import Context
from cudaparticlesfoam_amd.cases import pitzdaily as pz
from cudaparticlesfoam_amd.parallel import x_slab_renumbering
dev = torch.device("cuda", 0)
mesh0 = pz.pitzdaily_mesh(); c0, _ = mesh0.cell_centres_volumes()
mesh = mesh0.renumber_cells(x_slab_renumbering(c0))
n = 10_000_000
for D in (0.0, 1.5e-5):
    ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_mesh(mesh); ctx.set_velocity(pz.uniform_u(mesh))
    x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 1000, dev)
    g = torch.arange(n, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()
    ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
    ctx.set_option("stats", 1)
    ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, 1e-4, D, 0, 5, 0)
    torch.cuda.synchronize()
    a = ctx.counters()
    ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, 1e-4, D, 5, 1, 0)
    torch.cuda.synchronize()
    b = ctx.counters()
    d = {k: b[k] - a[k] for k in b}
    tiles = n / 64
    print("D", D, ctx.step_kernel_name(D, 0), d, "tile rounds/tile", (d["particle_steps"] - n) / tiles, "batch rounds/tile", d["reflections"] / tiles,
          "parked/tile", d["lost"] / tiles, flush=True)
    ctx.close()
