#!/usr/bin/env python3
"""Developer tool (GPU box): the UPPER BOUND of what any cure for visit-count divergence inside a tile can buy (round-4 verdict,
item 4).  The census says a tile of 64 particles takes as many rounds as its slowest lane (3.6 on the 3-D box) while the mean
visit count is ~2.  Here the cloud is put in the order an oracle would choose: the particles' ACTUAL visit counts of the coming
cycle (computed on the CPU by the checker, D = 0) decide the order inside windows of W consecutive particles of the sorted cloud
(a window keeps its cells, so the record cache sees the same cells).  Then the very same cycle is timed in both orders.  No
predictor can beat that order; if the gain is small the idea is dead.
  python tools/visit_order_bound.py [--case box3d --field swirl --particles 1e7] [--windows 256,1024,4096]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="box3d"); ap.add_argument("--field", default="swirl")
    ap.add_argument("--particles", type=float, default=1e7); ap.add_argument("--windows", default="64,256,1024,4096,65536")
    ap.add_argument("--pre-steps", type=int, default=5); ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    import torch
    from _cases import POLY_CASES, POLY_DT, make_case
    from _spinup import device_spinup
    from cudaparticlesfoam_amd.api import Context
    from oracle import oracle as O
    dev = torch.device("cuda", 0)
    ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n = int(a.particles)
    mesh, x, y, z, c, fields = make_case(a.case, ctx, torch, n, dev, a.field)
    U = fields[a.field] if a.field in fields else next(iter(fields.values()))
    ctx.set_velocity(U)
    dt = POLY_DT if a.case in POLY_CASES else 1e-4
    g = torch.arange(n, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
    ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, dt, 0.0, 0, a.pre_steps, 0)
    ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)            # the state a step of the bench starts from: freshly sorted
    torch.cuda.synchronize()
    state = [t.clone() for t in (x, y, z, c)]
    # the oracle's answer: visits of the coming cycle
    O.build(); cw = O.CellWalk(); tab = cw.build(mesh)
    hx, hy, hz, hc = (t.cpu().numpy().copy() for t in state)
    visits, refl = cw.step_count(hx, hy, hz, hc, dt, tab, U, nthreads=cw.hw_threads)
    live = state[3].cpu().numpy() >= 0
    tiles = (n + 63) // 64
    vt = np.zeros(tiles * 64, np.int32); vt[:n] = np.where(live, visits + refl, 0)
    vt = vt.reshape(tiles, 64)
    rounds_now = float(vt.max(1).mean()); mean_visits = float(vt[vt > 0].mean())

    def timed(arrs):
        device_spinup(ctx, torch, *[t.clone() for t in arrs], n, dt)
        tot = 0.0
        for _ in range(a.reps):
            w = [t.clone() for t in arrs]
            torch.cuda.synchronize()
            ctx.timing_enable(True); ctx.timing_read()
            ctx.step_dev(p(w[0]), p(w[1]), p(w[2]), p(w[3]), None, None, n, dt, 0.0, a.pre_steps, 1, 0)
            l, ms = ctx.timing_read(); ctx.timing_enable(False)
            tot += ms / max(l, 1)
        return tot / a.reps

    base = timed(state)
    rows = []
    key = torch.from_numpy((visits + refl).astype(np.int64)).to(dev)
    for W in [int(v) for v in a.windows.split(",")]:
        win = torch.arange(n, device=dev) // W
        order = torch.argsort(win * 1024 + key, stable=True)           # inside a window: by visit count, else as sorted
        arrs = [t[order].contiguous() for t in state]
        v2 = np.zeros(tiles * 64, np.int32); v2[:n] = np.where(live, visits + refl, 0)[order.cpu().numpy()]
        rounds = float(v2.reshape(tiles, 64).max(1).mean())
        ms = timed(arrs)
        rows.append({"window": W, "kernel_ms": round(ms, 4), "gain": round(1 - ms / base, 4), "rounds_per_tile_bound": round(rounds, 3)})
    # the oracle of the "exit-time bins" predictor: cell-major order kept, the particles of ONE CELL ordered by their actual visit count
    # (instead of by their sub-box): tiles keep their cells, lanes with equal counts sit together where a cell holds enough particles
    cellkey = state[3].to(torch.int64) & 0xFFFFFFFF
    order = torch.argsort(cellkey * 1024 + key, stable=True)
    arrs = [t[order].contiguous() for t in state]
    v2 = np.zeros(tiles * 64, np.int32); v2[:n] = np.where(live, visits + refl, 0)[order.cpu().numpy()]
    ms = timed(arrs)
    rows.append({"window": "per cell", "kernel_ms": round(ms, 4), "gain": round(1 - ms / base, 4),
                 "rounds_per_tile_bound": round(float(v2.reshape(tiles, 64).max(1).mean()), 3)})
    print(json.dumps({"case": a.case, "field": a.field, "particles": n, "cells": mesh.n_cells, "kernel": ctx.step_kernel_name(0.0, 0),
                      "kernel_ms_sorted": round(base, 4), "mean_visits": round(mean_visits, 3), "rounds_per_tile_bound_sorted": round(rounds_now, 3),
                      "ordered_by_actual_visits": rows}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
