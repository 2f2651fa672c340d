import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cudaparticlesfoam_amd.api import Context
from cudaparticlesfoam_amd.cases import refined_box
from oracle import oracle as O
O.build()
mesh, _ = refined_box(8, 6, 5, (0, 0, 0), (8, 6, 5), ((2.0, 1.5, 1.0), (6.0, 4.5, 4.0)), grading=(2.0, 1.0, 0.5))
cw = O.CellWalk(); t = cw.build(mesh)
rng = np.random.default_rng(23)
n = 200_000
xyz = rng.uniform([0, 0, 0], [8, 6, 5], size=(n, 3))
U = rng.normal(size=(mesh.n_cells, 3)) * 0.5
res = {}
for name, opts in (("stream2", {}), ("generic", {"step_variant": 0}), ("stream2_again", {})):
    for cycles in (1, 5, 60):
        ctx = Context(0)
        for k, v in opts.items(): ctx.set_option(k, v)
        ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz); ctx.locate_initial(); ctx.sort_by_cell()
        ctx.step(0.05, 0.4, cycles)
        xyzw, cell = ctx.get_particles()
        res[(name, cycles)] = (xyzw.copy(), cell.copy())
        nf = np.diff(t.cell_off); pl = t.planes.reshape(-1, 4)
        worst = np.zeros(n)
        ok = cell >= 0
        for k in range(int(nf.max())):
            has = ok & (nf[np.maximum(cell, 0)] > k)
            p = pl[t.cell_off[cell[has]] + k]
            fd = p[:, 3] - (p[:, :3] * xyzw[has, :3]).sum(1)
            worst[has] = np.maximum(worst[has], fd)
        bad = np.nonzero(worst > 1e-9)[0]
        print(name, cycles, ctx.step_kernel_name(0.4, 0), "lost", int((cell < 0).sum()), "violators", bad.size, "worst", float(worst.max()),
              "nf of violators' cells", np.bincount(nf[cell[bad]])[6:] if bad.size else None, flush=True)
        ctx.close()
for cycles in (1, 5, 60):
    a, b = res[("stream2", cycles)], res[("generic", cycles)]
    d = np.nonzero((a[1] != b[1]) | (a[0][:, :3] != b[0][:, :3]).any(1))[0]
    print("cycles", cycles, "stream2 vs generic differ:", d.size, "stream2 vs stream2_again:", int(((res[("stream2", cycles)][0] != res[("stream2_again", cycles)][0]).any(1)).sum()))
    if d.size:
        i = d[0]; print("  first", i, a[0][i], a[1][i], b[0][i], b[1][i], "start", xyz[i])
