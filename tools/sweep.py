#!/usr/bin/env python3
"""Developer tool (GPU box): times the step-kernel variants on the bench workload and checks that every
variant gives bit-identical particles.  python tools/sweep.py [--particles 1e7] [--steps 30] [--field uniform]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=float, default=1e7)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--field", default="uniform")
    ap.add_argument("--variants", default="0,1,2,3")
    ap.add_argument("--unsorted", action="store_true")
    ap.add_argument("--no-stats", action="store_true")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--D", type=float, default=0.0, help="diffusion coefficient (Brownian kick on when > 0)")
    ap.add_argument("--store-vel", action="store_true", help="also store the velocity per particle (output cycles)")
    ap.add_argument("--opt", action="append", default=[], help="cpf_set_option key=value (repeatable), e.g. stream_tiles_per_chunk=8")
    ap.add_argument("--label", default="")
    ap.add_argument("--spinup-ms", type=float, default=100.0)
    ap.add_argument("--no-floor", action="store_true", help="skip the zero-cycle launches (they would mix into PMC means)")
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
    ctx.set_mesh(mesh); ctx.set_velocity(U)
    n = int(args.particles)
    x0, y0, z0, c0 = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 1000, dev)
    g0 = torch.arange(n, dtype=torch.int64, device=dev)
    for kv in args.opt:                      # (before the sort: some options shape the sort key)
        k, v = kv.split("=")
        ctx.set_option(k, float(v))
    if not args.unsorted:
        ctx.sort_by_cell_dev(x0.data_ptr(), y0.data_ptr(), z0.data_ptr(), c0.data_ptr(), g0.data_ptr(), n)
    torch.cuda.synchronize()
    ref = None
    rows = []
    if args.no_stats:
        ctx.set_option("stats", 0)
    for v in [int(s) for s in args.variants.split(",")]:
        ctx.set_option("step_variant", v)
        x, y, z, c = x0.clone(), y0.clone(), z0.clone(), c0.clone()
        p = lambda t: t.data_ptr()   # noqa: E731
        vel = torch.empty(3 * n, dtype=torch.float64, device=dev) if args.store_vel else None
        fl = args.flags | (2 if args.store_vel else 0)
        pv = None if vel is None else vel.data_ptr()
        if args.spinup_ms > 0:                   # steady device clocks (tools/_spinup.py); PMC passes switch it off
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from _spinup import device_spinup
            stats_on = not args.no_stats
            device_spinup(ctx, torch, x, y, z, c, n, 1e-4, args.spinup_ms)
            ctx.set_option("stats", 1 if stats_on else 0)
        ctx.step_dev(p(x), p(y), p(z), p(c), None, pv, n, 1e-4, args.D, 0, args.warmup, fl)
        torch.cuda.synchronize()
        ctx.timing_enable(True)
        ctx.step_dev(p(x), p(y), p(z), p(c), None, pv, n, 1e-4, args.D, args.warmup, args.steps, fl)
        launches, ms = ctx.timing_read()
        ctx.timing_enable(False)
        avg = ms / launches
        same = None
        if ref is None:
            ref = (x, y, z, c)
        else:
            same = bool(torch.equal(x, ref[0]) and torch.equal(y, ref[1]) and torch.equal(z, ref[2]) and torch.equal(c, ref[3]))
        rows.append(dict(label=args.label, opts=args.opt, variant=v, kernel_ms=round(avg, 4), gps=round(n / avg / 1e6, 2), gbs=round(56 * n / avg / 1e6, 1),
                         identical_to_first=same))
        print(json.dumps(rows[-1]), flush=True)
    if args.no_floor:
        ctx.close()
        return
    # IO floor: same loads/stores, zero cycles (no walk)
    from cudaparticlesfoam_amd import _lib as L
    x, y, z, c = x0.clone(), y0.clone(), z0.clone(), c0.clone()
    torch.cuda.synchronize()
    ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, 1e-4, 0.0, 0, 0, L.STEP_FUSE_CYCLES)
    ctx.timing_enable(True)
    for _ in range(20):
        ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, 1e-4, 0.0, 0, 0, L.STEP_FUSE_CYCLES)
    launches, ms = ctx.timing_read()
    print(json.dumps(dict(variant="zero-cycle (IO floor)", kernel_ms=round(ms / launches, 4),
                          gbs=round(56 * n / (ms / launches) / 1e6, 1))), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
