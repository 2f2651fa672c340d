#!/usr/bin/env python3
"""Secondary measurement (BASELINE.json configs[4] shape, single GPU): a TJunction/"motorBike-scale" mesh
(pitzDaily refined r x r in-plane: r=6 -> 440 100 hex cells, 113 MB of cell records: no longer L2-resident),
1e7 particles, U(t) re-uploaded from the host before every Eulerian step, 10 Lagrangian sub-cycles each.
Reports upload time separately.  python tools/bench_pimple.py [--refine 6]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--refine", type=int, default=6)
    ap.add_argument("--particles", type=float, default=1e7)
    ap.add_argument("--eulerian-steps", type=int, default=10)
    ap.add_argument("--cycles", type=int, default=10)
    a = ap.parse_args()
    import torch
    import bench
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.cases import pitzdaily as pz
    from cudaparticlesfoam_amd.parallel import x_slab_renumbering
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    m0 = pz.pitzdaily_mesh(refine=a.refine); c0, _ = m0.cell_centres_volumes()
    mesh = m0.renumber_cells(x_slab_renumbering(c0)); cen, _ = mesh.cell_centres_volumes()
    t_mesh = time.perf_counter() - t0
    ctx = Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for kv in os.environ.get("CPF_OPTS", "").split():              # e.g. CPF_OPTS="flat_walk=0"
        ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    t0 = time.perf_counter(); ctx.set_mesh(mesh); t_ingest = time.perf_counter() - t0
    base = pz.analytic_step_u(mesh, cen)
    ctx.set_velocity(base); ctx.set_option("stats", 0)
    n = int(a.particles)
    x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 5, dev)
    g = torch.arange(n, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    ctx.sort_by_cell_dev(p(x), p(y), p(z), p(c), p(g), n)
    dt = 1e-4 / a.refine
    ctx.step_dev(p(x), p(y), p(z), p(c), p(g), None, n, dt, 0.0, 0, 5, 0); torch.cuda.synchronize()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from _spinup import device_spinup
    device_spinup(ctx, torch, x, y, z, c, n, dt)
    up = 0.0; ctx.timing_enable(True)
    torch.cuda.synchronize(); w0 = time.perf_counter()
    for e in range(a.eulerian_steps):
        U = base * (1.0 + 0.3 * np.sin(0.7 * e))
        torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.set_velocity(U); up += time.perf_counter() - t0
        ctx.step_dev(p(x), p(y), p(z), p(c), p(g), None, n, dt, 0.0, 5 + e * a.cycles, a.cycles, 0)
    torch.cuda.synchronize(); wall = time.perf_counter() - w0
    launches, ms = ctx.timing_read()
    print(json.dumps(dict(cells=mesh.n_cells, mesh_device_MB=round(ctx.mesh_info()["device_bytes"] / 1e6, 1),
                          python_mesher_s=round(t_mesh, 2), set_mesh_ingest_s=round(t_ingest, 3), particles=n,
                          kernel_ms=round(ms / launches, 4), Gparticle_steps_per_s=round(n / (ms / launches) / 1e6, 2),
                          roofline_GBs=round(56 * n / (ms / launches) / 1e6, 1),
                          U_upload_ms_per_eulerian_step=round(up / a.eulerian_steps * 1e3, 3),
                          U_bytes_per_refresh=mesh.n_cells * 24, reference_bytes_per_refresh=mesh.n_cells * 12 * 24,
                          wall_ms_per_lagrangian_cycle=round(wall / (a.eulerian_steps * a.cycles) * 1e3, 4))))
    ctx.close()


if __name__ == "__main__":
    main()
