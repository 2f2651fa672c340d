#!/usr/bin/env python3
"""Developer tool (GPU box): time of one re-sort of a cloud that was sorted `--age` steps ago with each key sort ("sort_method"
0 = library, 2 = this library's own, the default), on pitzDaily (1e7 particles, 21 key bits) and TJunction (4e6, 24).
  python tools/sort_timing.py [--age 25] [--D 1.5e-5]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from _cases import make_case
from cudaparticlesfoam_amd.api import Context


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--age", type=int, default=25); ap.add_argument("--D", type=float, default=1.5e-5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    for case, n in (("pitz", 10_000_000), ("tjunction", 4_000_000)):
        ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        mesh, x, y, z, c, fields = make_case(case, ctx, torch, n, dev, None)
        ctx.set_velocity(list(fields.values())[-1])
        g = torch.arange(n, dtype=torch.int64, device=dev)
        p = lambda t: t.data_ptr()   # noqa: E731
        ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
        ctx.step_dev(p(x), p(y), p(z), p(c), p(g), None, n, 1e-4, a.D, 0, a.age, 0)
        torch.cuda.synchronize()
        out = [torch.empty_like(t) for t in (x, y, z, c, g)]
        row = {"case": case, "particles": n, "cells": mesh.n_cells, "age_steps": a.age, "D": a.D}
        for method in (0, 2, 0, 2):
            ctx.set_option("sort_method", method)
            ctx.sort_by_cell_dev_to(p(x), p(y), p(z), p(c), p(g), *(p(t) for t in out), n); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                ctx.sort_by_cell_dev_to(p(x), p(y), p(z), p(c), p(g), *(p(t) for t in out), n)
            torch.cuda.synchronize()
            row.setdefault("ms_method%d" % method, []).append(round((time.perf_counter() - t0) / 10 * 1e3, 4))
        print(json.dumps(row), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
