#!/usr/bin/env python3
"""Developer tool (GPU box): the streaming step kernel (step_variant 4) against the wave-cooperative one (3) --
bit-identical particles for every instantiation (Brownian / no-reflect / stored velocity / statistics / fused
cycles), ragged cloud sizes, unsorted clouds, chunk lengths; then timings.  python tools/stream_check.py"""
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.cases import pitzdaily as pz
    from cudaparticlesfoam_amd.parallel import x_slab_renumbering
    dev = torch.device("cuda", 0)
    mesh0 = pz.pitzdaily_mesh(); c0, _ = mesh0.cell_centres_volumes()
    mesh = mesh0.renumber_cells(x_slab_renumbering(c0))
    centres, _ = mesh.cell_centres_volumes()
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_mesh(mesh)
    p = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    bad = 0
    for field, n, sort in [("analytic", 200_003, True), ("uniform", 100_000, False), ("analytic", 1, True),
                           ("analytic", 63, True), ("analytic", 64, True), ("analytic", 65, True),
                           ("analytic", 4097, False), ("uniform", 1_000_000, True)]:
        U = pz.uniform_u(mesh) if field == "uniform" else pz.analytic_step_u(mesh, centres)
        ctx.set_velocity(U)
        x0, y0, z0, c0_ = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 77 + n, dev)
        g0 = torch.arange(n, dtype=torch.int64, device=dev)
        # some lost / frozen particles in the mix
        if n > 100:
            c0_[5] = -1; c0_[17] = -2; c0_[n - 3] = -1
        if sort:
            ctx.sort_by_cell_dev(p(x0), p(y0), p(z0), p(c0_), p(g0), n)
        for D, noref, sv, stats, fused, (tpc, il) in itertools.product((0.0, 1.5e-5), (0, 1), (0, 1), (0, 1), (0, 1), ((1, 0.2), (4, 0.2), (16, 0.0), (4, 0.7))):
            if n > 300_000 and (sv or stats or noref or tpc == 1):
                continue
            flags = (L.STEP_NO_REFLECT if noref else 0) | (L.STEP_STORE_VEL if sv else 0) | (L.STEP_FUSE_CYCLES if fused else 0)
            outs = []
            if os.environ.get("CPF_CHECK_VERBOSE"):
                print("combo", dict(field=field, n=n, D=D, noref=noref, sv=sv, stats=stats, fused=fused, tpc=tpc, il=il), flush=True)
            for variant in (3, int(os.environ.get("CPF_CHECK_VARIANT", "4"))):
                ctx.set_option("step_variant", variant)
                ctx.set_option("stats", stats)
                ctx.set_option("stream_tiles_per_chunk", tpc)
                ctx.set_option("stream_tail_fraction", il)
                if os.environ.get("CPF_CHECK_LOOKUP"):
                    ctx.set_option("stream_lookup", int(os.environ["CPF_CHECK_LOOKUP"]))
                x, y, z, c = x0.clone(), y0.clone(), z0.clone(), c0_.clone()
                vel = torch.zeros(3 * n, dtype=torch.float64, device=dev) if sv else None
                cnt0 = ctx.counters()
                ctx.step_dev(p(x), p(y), p(z), p(c), p(g0), p(vel), n, 1e-4, D, 3, 7, flags)
                torch.cuda.synchronize()
                cnt1 = ctx.counters()
                outs.append((x, y, z, c, vel, {k: cnt1[k] - cnt0[k] for k in cnt1}))
            a, b = outs
            same = all(torch.equal(a[i], b[i]) for i in range(4)) and (vel is None or torch.equal(a[4], b[4])) and a[5] == b[5]
            if not same:
                bad += 1
                nd = int((a[3] != b[3]).sum()); nx = int((a[0] != b[0]).sum())
                ny = int((a[1] != b[1]).sum()); nz = int((a[2] != b[2]).sum())
                nv = int((a[4] != b[4]).sum()) if vel is not None else 0
                print("  y_differ", ny, "z_differ", nz, "vel_differ", nv, "max|dz|", float((a[2] - b[2]).abs().max()), flush=True)
                print("MISMATCH", dict(field=field, n=n, sort=sort, D=D, noref=noref, sv=sv, stats=stats, fused=fused,
                                       tpc=tpc, il=il, cells_differ=nd, x_differ=nx, cnt3=a[5], cnt4=b[5]), flush=True)
        print("checked", field, n, "sorted" if sort else "unsorted", "bad so far:", bad, flush=True)
    ctx.set_option("stats", 0)
    print("RESULT", "ALL IDENTICAL" if bad == 0 else "%d MISMATCHES" % bad, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
