#!/usr/bin/env python3
"""Developer tool (GPU box): how fast does the step kernel slow down as a cell-sorted cloud disorders,
and what does a re-sort cost?  python tools/drift.py [--field analytic] [--windows 8] [--steps 50]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--field", default="analytic")
    ap.add_argument("--particles", type=float, default=1e7)
    ap.add_argument("--windows", type=int, default=8)
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    import torch
    import bench
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.cases import pitzdaily as pz
    from cudaparticlesfoam_amd.parallel import x_slab_renumbering
    dev = torch.device("cuda", 0)
    mesh0 = pz.pitzdaily_mesh(); c0, _ = mesh0.cell_centres_volumes()
    mesh = mesh0.renumber_cells(x_slab_renumbering(c0))
    centres, _ = mesh.cell_centres_volumes()
    U = pz.uniform_u(mesh) if args.field == "uniform" else pz.analytic_step_u(mesh, centres)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_option("stats", 0)
    n = int(args.particles)
    x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 1000, dev)
    g = torch.arange(n, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731

    def sort_ms():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3

    print(json.dumps(dict(event="initial sort of a random cloud", ms=round(sort_ms(), 3))), flush=True)
    step = 0
    for w in range(args.windows):
        ctx.timing_enable(True)
        ctx.step_dev(p(x), p(y), p(z), p(c), p(g), None, n, 1e-4, 0.0, step, args.steps, 0)
        launches, ms = ctx.timing_read(); ctx.timing_enable(False)
        step += args.steps
        print(json.dumps(dict(window=w, steps_done=step, kernel_ms=round(ms / launches, 4))), flush=True)
    print(json.dumps(dict(event="re-sort after %d steps" % step, ms=round(sort_ms(), 3))), flush=True)
    ctx.timing_enable(True)
    ctx.step_dev(p(x), p(y), p(z), p(c), p(g), None, n, 1e-4, 0.0, step, 20, 0)
    launches, ms = ctx.timing_read()
    print(json.dumps(dict(event="after re-sort", kernel_ms=round(ms / launches, 4))), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
