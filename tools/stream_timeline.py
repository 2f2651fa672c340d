#!/usr/bin/env python3
"""Developer tool (GPU box, library built with -DCPF_STREAM_TIMELINE): when does each persistent wave of the streaming
step kernel start and end, how many tiles and rounds did it do.  python tools/stream_timeline.py [--opt k=v ...]"""
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
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--label", default="")
    ap.add_argument("--pre-steps", type=int, default=5, help="steps after the sort and before the measured launch")
    args = ap.parse_args()
    import torch
    import bench
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.cases import pitzdaily as pz
    from cudaparticlesfoam_amd.parallel import x_slab_renumbering
    dev = torch.device("cuda", 0)
    mesh0 = pz.pitzdaily_mesh(); c0, _ = mesh0.cell_centres_volumes()
    mesh = mesh0.renumber_cells(x_slab_renumbering(c0))
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_mesh(mesh); ctx.set_velocity(pz.uniform_u(mesh))
    n = int(args.particles)
    x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 1000, dev)
    g = torch.arange(n, dtype=torch.int64, device=dev)
    ctx.set_option("stats", 0); ctx.set_option("step_variant", 4)
    for kv in args.opt:
        k, v = kv.split("="); ctx.set_option(k, float(v))
    ctx.sort_by_cell_dev(x.data_ptr(), y.data_ptr(), z.data_ptr(), c.data_ptr(), g.data_ptr(), n)
    tl = torch.zeros(4 * 16384, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    for _ in range(args.pre_steps):
        ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, 1e-4, 0.0, 0, 1, 0)
    torch.cuda.synchronize()
    tl.zero_()
    ctx.timing_enable(True)
    ctx.step_dev(p(x), p(y), p(z), p(c), None, p(tl), n, 1e-4, 0.0, 5, 1, 0)
    launches, ms = ctx.timing_read()
    t = tl.cpu().numpy().reshape(-1, 4)
    t = t[t[:, 1] > 0]
    t0, t1 = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64)
    base = t0.min()
    start = (t0 - base) * 10.0 / 1e3          # us (100 MHz ticks)
    end = (t1 - base) * 10.0 / 1e3
    q = lambda a: [round(float(v), 2) for v in np.percentile(a, [0, 5, 25, 50, 75, 95, 100])]   # noqa: E731
    out = dict(label=args.label, opts=args.opt, kernel_ms=round(ms / launches, 4), waves=int(t.shape[0]),
               start_us_pct=q(start), end_us_pct=q(end), life_us_pct=q(end - start),
               tiles_pct=q(t[:, 2]), rounds_per_tile=round(float(t[:, 3].sum() / max(1, t[:, 2].sum())), 3),
               us_per_tile_pct=q((end - start) / np.maximum(t[:, 2], 1)),
               slot_busy_fraction=round(float((end - start).sum() / (t.shape[0] * end.max())), 3))
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
