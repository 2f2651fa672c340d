#!/usr/bin/env python3
"""Developer tool (GPU box): time of the hand-off's split (cpf_pack_leavers_dev: count, scan, write, fill, mark) by the share of the
shard that leaves -- one context, 1.25e7 particles on pitzDaily, rank 0 of 2 with the cut placed so that the wanted share of the
particles lies beyond it -- and of the unpack of as many arrivals.
  python tools/pack_timing.py [--particles 1.25e7] [--fractions 0.003,0.03,0.1,0.5,0.9]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=float, default=1.25e7); ap.add_argument("--fractions", default="0.003,0.03,0.1,0.5,0.9")
    ap.add_argument("--ranks", type=int, default=8)
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.cases import pitzdaily as pz
    from cudaparticlesfoam_amd.parallel import x_slab_renumbering
    dev = torch.device("cuda", 0)
    m0 = pz.pitzdaily_mesh(); c0, _ = m0.cell_centres_volumes(); mesh = m0.renumber_cells(x_slab_renumbering(c0))
    ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream); ctx.set_mesh(mesh); ctx.set_velocity(pz.uniform_u(mesh))
    n = int(a.particles)
    x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 1, dev)
    g = torch.arange(n, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
    cs = torch.sort(c).values
    W = a.ranks
    sendbuf = torch.empty(n * L.HANDOFF_DOUBLES, dtype=torch.float64, device=dev)
    counts = torch.zeros(W, dtype=torch.int64, device=dev); nstay = torch.zeros(1, dtype=torch.int64, device=dev)
    for f in [float(v) for v in a.fractions.split(",")]:
        # rank 0 of W keeps the cells below the cut; the rest is spread evenly over the other ranks' ranges
        cut = int(cs[int(n * (1.0 - f))].item())
        lo = np.linspace(cut, mesh.n_cells, W).astype(np.int32)
        cell_lo = torch.from_numpy(np.concatenate([[0], lo]).astype(np.int32)).to(dev)
        ts = []
        for rep in range(4):
            w = [t.clone() for t in (x, y, z, c, g)]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.pack_leavers_dev(*[p(t) for t in w], n, p(cell_lo), W, 0, p(sendbuf), n, p(counts), p(nstay))
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        leavers = int(counts.sum().item())
        t1 = time.perf_counter()
        ctx.unpack_arrivals_dev(*[p(t) for t in w], int(nstay.item()), p(sendbuf), leavers)
        torch.cuda.synchronize(); tu = (time.perf_counter() - t1) * 1e3
        print(json.dumps({"particles": n, "ranks": W, "leavers": leavers, "share": round(leavers / n, 4), "split_ms": round(min(ts[1:]), 3),
                          "unpack_ms": round(tu, 3), "one_cycle_ms_for_scale": 0.14}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
