#!/usr/bin/env python3
"""Developer tool (GPU box): how fast does the step kernel slow down as the cloud's order goes stale?  One sort, then
windows of 50 steps without a re-sort (kernel ms from HIP events on every 4th launch), for the bench's uniform field and
for the sheared analytic one; then the cost of one sort.  python tools/sort_decay.py [windows]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch
    import bench
    from _spinup import device_spinup
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.cases import pitzdaily as pz
    from cudaparticlesfoam_amd.parallel import x_slab_renumbering
    windows = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    D = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0          # diffusion coefficient of the measured steps
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 50             # steps per window
    opts = [kv.split("=") for kv in sys.argv[4:]]                 # context options, e.g. sort_key_bits=110
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 10_000_000
    mesh0 = pz.pitzdaily_mesh(); c0, _ = mesh0.cell_centres_volumes()
    mesh = mesh0.renumber_cells(x_slab_renumbering(c0)); centres, _ = mesh.cell_centres_volumes()
    ctx.set_mesh(mesh)
    p = lambda t: t.data_ptr()   # noqa: E731
    ctx.set_option("timing_stride", 4)
    for k_, v_ in opts:
        ctx.set_option(k_, float(v_))
    for name, U in (("uniform", pz.uniform_u(mesh)), ("analytic", pz.analytic_step_u(mesh, centres))):
        ctx.set_velocity(U)
        x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 1000, dev)
        g = torch.arange(n, dtype=torch.int64, device=dev)
        ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
        device_spinup(ctx, torch, x, y, z, c, n, 1e-4)
        step, rows = 0, []
        for w in range(windows):
            ctx.timing_enable(True); ctx.timing_read()
            for _ in range(W):
                ctx.step_dev(p(x), p(y), p(z), p(c), p(g) if D > 0 else None, None, n, 1e-4, D, step, 1, 0); step += 1
            launches, ms = ctx.timing_read(); ctx.timing_enable(False)
            rows.append(round(ms / max(1, launches), 4))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
        torch.cuda.synchronize(); sort_ms = (time.perf_counter() - t0) / 5 * 1e3
        ctx.timing_enable(True); ctx.timing_read()
        for _ in range(W):
            ctx.step_dev(p(x), p(y), p(z), p(c), p(g) if D > 0 else None, None, n, 1e-4, D, step, 1, 0); step += 1
        launches, ms = ctx.timing_read(); ctx.timing_enable(False)
        print(json.dumps(dict(field=name, opts=sys.argv[4:], D=D, steps_per_window=W, kernel_ms_per_window_since_the_sort=rows, sort_ms=round(sort_ms, 3),
                              kernel_ms_right_after_a_new_sort=round(ms / max(1, launches), 4))), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
