#!/usr/bin/env python3
"""Developer tool (GPU box): the step kernel on one of the synthetic cases of tools/_cases.py -- sort, 5 steps with the
statistics on (visits per particle-step), device spin-up, then 20 timed steps (dispatch time stamps of every 4th launch).
Reports the rate on the particle bytes alone (56 B per particle-step, + 8 with the kick) AND, for meshes whose records do
not fit the L2 (4 MB per XCD), with the records streamed once per launch added (`records_bytes_once`: 256 B x cells --
SURVEY.md 8d: "add per visited cell ..." for meshes beyond the cache).
  python tools/bench_case.py --case box3d [--field swirl] [--particles 1e7] [--D 0] [--opt k=v ...] [--steps 20] [--fused]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="box3d"); ap.add_argument("--field", default=None, help="one field of the case (default: all)")
    ap.add_argument("--particles", type=float, default=1e7); ap.add_argument("--D", type=float, default=0.0)
    ap.add_argument("--steps", type=int, default=20); ap.add_argument("--pre-steps", type=int, default=5)
    ap.add_argument("--opt", action="append", default=[]); ap.add_argument("--label", default="")
    ap.add_argument("--fused", action="store_true", help="the timed steps as ONE fused launch (CPF_STEP_FUSE_CYCLES)")
    ap.add_argument("--no-sort", action="store_true")
    a = ap.parse_args()
    import torch
    from _cases import POLY_CASES, POLY_DT, make_case
    from _spinup import device_spinup
    from cudaparticlesfoam_amd.api import Context
    dev = torch.device("cuda", 0)
    ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = int(a.particles)
    for kv in a.opt:
        if kv.split("=")[0] in ("mixed_records", "box_records"):          # (options that shape the mesh tables go in before the mesh)
            ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    mesh, x, y, z, c, fields = make_case(a.case, ctx, torch, n, dev, a.field)
    dt = POLY_DT if a.case in POLY_CASES else 1e-4
    for kv in a.opt:
        k_, v_ = kv.split("="); ctx.set_option(k_, float(v_))
    g = torch.arange(n, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    x0, y0, z0, c0 = x.clone(), y.clone(), z.clone(), c.clone()
    for name, U in fields.items():
        if a.field and name != a.field:
            continue
        ctx.set_velocity(U)
        x.copy_(x0); y.copy_(y0); z.copy_(z0); c.copy_(c0); g.copy_(torch.arange(n, dtype=torch.int64, device=dev))
        if not a.no_sort:
            ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
        pg = p(g) if a.D > 0 else None
        ctx.set_option("stats", 1); s0 = ctx.counters()
        ctx.step_dev(p(x), p(y), p(z), p(c), pg, None, n, dt, a.D, 0, a.pre_steps, 0)
        torch.cuda.synchronize(); s1 = ctx.counters(); ctx.set_option("stats", 0)
        device_spinup(ctx, torch, x, y, z, c, n, dt)
        ctx.timing_enable(True); ctx.timing_read()
        ctx.step_dev(p(x), p(y), p(z), p(c), pg, None, n, dt, a.D, a.pre_steps, a.steps, 4 if a.fused else 0)
        launches, ms = ctx.timing_read(); ctx.timing_enable(False)
        k = ms / max(launches, 1) / (a.steps if a.fused else 1)
        per = 56 + (8 if a.D > 0 else 0)
        rec_once = (128 if ctx.step_kernel_name(a.D, 0).endswith(", 6>") else 256) * mesh.n_cells      # (box records: 128 B)
        print(json.dumps(dict(label=a.label, case=a.case, field=name, opts=a.opt, D=a.D, cells=mesh.n_cells, particles=n,
                              kernel=ctx.step_kernel_name(a.D, 0), kernel_ms=round(k, 4),
                              Gparticle_steps_per_s=round(n / k / 1e6, 2), roofline_GBs=round(per * n / k / 1e6, 1),
                              frac=round(per * n / k / 1e6 / 8000.0, 4), records_bytes_once=rec_once,
                              frac_with_records_once=round((per * n + rec_once) / k / 1e6 / 8000.0, 4),
                              visits_per_particle_step=round((s1["cells_visited"] - s0["cells_visited"]) /
                                                             max(1, s1["particle_steps"] - s0["particle_steps"]), 3))), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
