#!/usr/bin/env python3
"""Developer tool (GPU box): is the step kernel's time in the first hundred steps after seeding a property of the
workload (visits per particle-step drifting as the cloud advects) or of the device (clocks after an idle phase)?
Windows of 100 steps: kernel ms (HIP events on every 4th launch), visits per particle-step of one statistics step at the
window's end, the same again after the device sat idle for 3 s."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.cases import pitzdaily as pz
    from cudaparticlesfoam_amd.parallel import x_slab_renumbering
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = 10_000_000
    mesh0 = pz.pitzdaily_mesh(); c0, _ = mesh0.cell_centres_volumes()
    mesh = mesh0.renumber_cells(x_slab_renumbering(c0))
    ctx.set_mesh(mesh); ctx.set_velocity(pz.uniform_u(mesh))
    x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 1000, dev)
    g = torch.arange(n, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    ctx.set_option("timing_stride", 4)
    step = 0

    def window(label, k=100):
        nonlocal step
        ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
        ctx.set_option("stats", 0)
        ctx.timing_enable(True); ctx.timing_read()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k):
            ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, 1e-4, 0.0, step, 1, 0); step += 1
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / k * 1e3
        launches, ms = ctx.timing_read(); ctx.timing_enable(False)
        ctx.set_option("stats", 1)
        a = ctx.counters()
        ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, 1e-4, 0.0, step, 1, 0); step += 1
        torch.cuda.synchronize()
        b = ctx.counters()
        print(json.dumps(dict(window=label, first_step=step - k - 1, wall_ms_per_step=round(wall, 4),
                              kernel_ms=round(ms / max(1, launches), 4),
                              visits=round((b["cells_visited"] - a["cells_visited"]) / max(1, b["particle_steps"] - a["particle_steps"]), 3),
                              reflections=round((b["reflections"] - a["reflections"]) / max(1, b["particle_steps"] - a["particle_steps"]), 4))),
              flush=True)

    for w in range(5):
        window("w%d" % w)
    time.sleep(3.0)
    window("after 3 s idle")
    window("next")
    # how long does the device take to reach its steady rate after an idle phase?  40 launches per sample
    time.sleep(3.0)
    ctx.set_option("stats", 0)
    prof = []
    t_start = time.perf_counter()
    for s_ in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, 1e-4, 0.0, step, 1, 0); step += 1
        torch.cuda.synchronize()
        prof.append((round((t0 - t_start) * 1e3, 1), round((time.perf_counter() - t0) / 40 * 1e3, 4)))
    print(json.dumps(dict(window="ramp after 3 s idle: (ms since start, wall ms per step) per 40 launches", samples=prof)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
