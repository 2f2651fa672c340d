"""CPU check (oracle only): after k cycles, is every particle inside the cell it claims?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from cudaparticlesfoam_amd.cases import refined_box, refined_pitzdaily
from oracle import oracle as O
O.build(); cw = O.CellWalk()

def worst_outside(t, xyz, cell):
    nf = np.diff(t.cell_off); pl = t.planes.reshape(-1, 4)
    worst = np.full(len(cell), -1e300); ok = cell >= 0
    for k in range(int(nf.max())):
        has = ok & (nf[np.maximum(cell, 0)] > k)
        p = pl[t.cell_off[cell[has]] + k]
        fd = p[:, 3] - (p[:, :3] * xyz[has]).sum(1)
        worst[has] = np.maximum(worst[has], fd)
    return worst

def run(name, mesh, lo, hi, dt, scale, D, cycles_list, n=200000):
    t = cw.build(mesh); rng = np.random.default_rng(23)
    xyz = rng.uniform(lo, hi, size=(n, 3)); U = rng.normal(size=(mesh.n_cells, 3)) * scale
    x, y, z = (xyz[:, k].copy() for k in range(3))
    cell = cw.locate_initial(x, y, z, t, nthreads=8)
    keep = cell >= 0
    x, y, z, cell = x[keep].copy(), y[keep].copy(), z[keep].copy(), cell[keep].copy()
    gid = np.arange(len(x), dtype=np.int64); done = 0
    for c in cycles_list:
        st = cw.step(x, y, z, cell, dt, c - done, t, U, nthreads=8, D=D, gid=gid, step0=done, seed=5); done = c
        w = worst_outside(t, np.stack([x, y, z], 1), cell)
        print(name, "D", D, "cycles", c, "lost", int((cell < 0).sum()), "violators(>1e-9)", int((w > 1e-9).sum()), "worst", float(w.max()), flush=True)

mesh = refined_box(8, 6, 5, (0, 0, 0), (8, 6, 5), ((2.0, 1.5, 1.0), (6.0, 4.5, 4.0)), grading=(2.0, 1.0, 0.5))[0]
for D in (0.0, 0.4):
    run("refined_box", mesh, [0, 0, 0], [8, 6, 5], 0.05, 0.5, D, (1, 5, 30, 100))

from cudaparticlesfoam_amd.cases import pitzdaily as pz
mesh, parent = refined_pitzdaily()
t = cw.build(mesh)
centres, _ = mesh.cell_centres_volumes()
U = pz.analytic_step_u(mesh, centres)
xyz = pz.uniform_points(99, 300_000, *pz.DOMAIN_BOX)
for D in (0.0, 1e-5):
    x, y, z = (xyz[:, k].copy() for k in range(3))
    cell = cw.locate_initial(x, y, z, t, nthreads=8); keep = cell >= 0
    x, y, z, cell = x[keep].copy(), y[keep].copy(), z[keep].copy(), cell[keep].copy()
    gid = np.arange(len(x), dtype=np.int64); done = 0
    for c in (1, 10, 50, 200):
        cw.step(x, y, z, cell, 1e-5, c - done, t, U, nthreads=8, D=D, gid=gid, step0=done, seed=5); done = c
        w = worst_outside(t, np.stack([x, y, z], 1), cell)
        print("refined_pitz D", D, "cycles", c, "n", len(x), "lost", int((cell < 0).sum()), "violators(>1e-9)", int((w > 1e-9).sum()), "worst", float(w.max()), flush=True)
