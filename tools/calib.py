#!/usr/bin/env python3
"""Developer tool (GPU box): launches the step kernel with ZERO cycles -- the same loads and stores as a real
step (read x,y,z,cell; write x,y,z,cell = 56 B/particle) and no walk -- so that the rocprofv3 FETCH_SIZE /
WRITE_SIZE counters can be calibrated on a known byte count in this kernel's own access pattern
(MI355X_MICROARCH.md "HBM": FETCH_SIZE under-reports some access widths on gfx950)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.api import Context
    from cudaparticlesfoam_amd.cases import pitzdaily as pz
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda", 0)
    mesh = pz.pitzdaily_mesh()
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_mesh(mesh); ctx.set_velocity(pz.uniform_u(mesh))
    x, y, z, c = bench.seed_in_fluid(ctx, torch, n, pz.DOMAIN_BOX, 1, dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    for _ in range(reps):
        ctx.step_dev(p(x), p(y), p(z), p(c), None, None, n, 1e-4, 0.0, 0, 0, L.STEP_FUSE_CYCLES)
    torch.cuda.synchronize()
    print("calib: %d launches of the zero-cycle step kernel, %d particles, %d algorithmic bytes each" % (reps, n, 56 * n))
    ctx.close()


if __name__ == "__main__":
    main()
