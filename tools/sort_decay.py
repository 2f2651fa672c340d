#!/usr/bin/env python3
"""Developer tool (GPU box): how fast does the step kernel slow down as the cloud's order goes stale?  One sort, then
windows of W steps without a re-sort (kernel ms from the dispatch time stamps of every 4th launch), then the cost of one
sort and the rate (sort + steps) / interval that every re-sort interval would give.
  python tools/sort_decay.py [--case pitz|box3d|tjunction] [--field NAME] [--windows 10] [--W 10] [--D 0] [--opt k=v ...]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="pitz"); ap.add_argument("--field", default=None)
    ap.add_argument("--windows", type=int, default=10); ap.add_argument("--W", type=int, default=10)
    ap.add_argument("--D", type=float, default=0.0); ap.add_argument("--particles", type=float, default=1e7)
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    import torch
    from _cases import make_case
    from _spinup import device_spinup
    from cudaparticlesfoam_amd.api import Context
    dev = torch.device("cuda", 0)
    ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = int(a.particles)
    mesh, x, y, z, c, fields = make_case(a.case, ctx, torch, n, dev, a.field)
    p = lambda t: t.data_ptr()   # noqa: E731
    ctx.set_option("timing_stride", 4); ctx.set_option("stats", 0)
    for kv in a.opt:
        k_, v_ = kv.split("="); ctx.set_option(k_, float(v_))
    g = torch.arange(n, dtype=torch.int64, device=dev)
    ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
    device_spinup(ctx, torch, x, y, z, c, n, 1e-4)
    step, rows = 0, []
    for w in range(a.windows):
        ctx.timing_enable(True); ctx.timing_read()
        ctx.step_dev(p(x), p(y), p(z), p(c), p(g) if a.D > 0 else None, None, n, 1e-4, a.D, step, a.W, 0); step += a.W
        launches, ms = ctx.timing_read(); ctx.timing_enable(False)
        rows.append(round(ms / max(1, launches), 4))
    o = [torch.empty_like(t) for t in (x, y, z, c, g)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        ctx.sort_by_cell_dev_to(p(x), p(y), p(z), p(c), p(g), *[p(t) for t in o], n)
        x, y, z, c, g, o = o[0], o[1], o[2], o[3], o[4], [x, y, z, c, g]
    torch.cuda.synchronize(); sort_ms = (time.perf_counter() - t0) / 4 * 1e3
    # (sort + the steps of one interval) / interval, for intervals of 1, 2, ... windows
    per_interval = {}
    for k in range(1, a.windows + 1):
        per_interval[k * a.W] = round((sort_ms + a.W * sum(rows[:k])) / (k * a.W), 4)
    best = min(per_interval, key=per_interval.get)
    print(json.dumps(dict(case=a.case, field=a.field, opts=a.opt, D=a.D, particles=n, cells=mesh.n_cells, steps_per_window=a.W,
                          kernel_ms_per_window_since_the_sort=rows, sort_ms=round(sort_ms, 3),
                          ms_per_step_all_in_by_sort_interval=per_interval, best_interval=best)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
