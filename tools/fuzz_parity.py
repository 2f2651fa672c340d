#!/usr/bin/env python3
"""Developer tool (GPU box): randomised differential campaign, HIP (through the C-ABI) vs the CPU cell-walk
statement, bit for bit.  Random graded/sheared hex blocks (tests/test_oracle_random._case), random cell-constant U,
time steps that cross several cells and bounce off several walls; every case runs with the statistics on and off
(two instantiations), plain and fused launches, sorted and unsorted clouds, and with exactly axis-aligned flow
(zero-denominator faces).  python tools/fuzz_parity.py [first_seed] [count] [mixed|poly|box|flat]
"flat": 2-D meshes extruded straight in z (1-3 layers of sheared, graded quads), fields without a z component, more than 128
particles per cell: the FLAT instantiation (csrc/cpf_walk.h "flat walk") against the CPU statement, positions compared as BITS.
"box": axis-aligned boxes (uniform or graded blockMesh boxes: BOX RECORDS, csrc/cpf_walk.h) with half of the cloud snapped to
fractions of the grid spacing and velocities that are +-1 / +-1/2 cells per step per axis on every other seed -- particles on
faces, edges and vertices, equal dT on two or three axes: the cases the three-candidate face test hands to the six-face form.
"mixed": a random subset of the block's cells is split 2 x 2 x 2 (2 x 2 x 1 for every third seed) first -- coarse cells
with one to six split faces, i.e. face groups in every combination -- and the CPU statement's result is also checked
against the domain's own invariant (nobody lost, everybody inside the cell they claim).
"poly": extruded polygon grids of random size with true polyhedra (cases/polygons.py: pentagonal, octagonal and dodecagonal
prisms -- 7, 10 and 14 planes: two-record cells and header records --, conformal or with hanging nodes), same invariant."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    mixed = len(sys.argv) > 3 and sys.argv[3] in ("mixed", "poly")
    poly = len(sys.argv) > 3 and sys.argv[3] == "poly"
    box = len(sys.argv) > 3 and sys.argv[3] == "box"
    flat = len(sys.argv) > 3 and sys.argv[3] == "flat"
    import torch  # noqa: F401  (its HIP runtime first)
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.api import Context
    from oracle import oracle as O
    from test_oracle_random import _case
    O.build()
    cw = O.CellWalk()
    bad = 0
    t0 = time.time()
    for seed in range(first, first + count):
        rng, mesh, U, dt = _case(seed)
        if poly:
            from cudaparticlesfoam_amd.cases import polygons as pg
            nx, ny, nz = int(rng.integers(4, 14)), int(rng.integers(4, 11)), int(rng.integers(1, 4))
            kind = seed % 5
            if kind == 0: mesh, _ = pg.cut_corner_box(nx, ny, nz, every=int(rng.integers(1, 6)), cut=float(rng.uniform(0.2, 0.45)))     # noqa: E701
            elif kind == 1: mesh, _ = pg.chamfered_box(nx, ny, nz, 1, cut=float(rng.uniform(0.15, 0.4)))                              # noqa: E701
            elif kind == 2: mesh, _ = pg.chamfered_box(nx, ny, nz, 2, cut=float(rng.uniform(0.15, 0.4)))                              # noqa: E701
            else: mesh, _ = pg.diamond_box(nx, ny, nz, period=int(rng.integers(1, 4)), cut=float(rng.uniform(0.15, 0.45)))            # noqa: E701
            U = rng.normal(size=(mesh.n_cells, 3)) * float(rng.choice([0.5, 1.5, 4.0])) + rng.normal(size=3)
            dt = float(rng.choice([0.05, 0.2, 0.5]))
        elif mixed:
            from cudaparticlesfoam_amd.cases.refine import refine_hexes
            mask = rng.random(mesh.n_cells) < rng.choice([0.1, 0.3, 0.5])
            mask[int(rng.integers(mesh.n_cells))] = True
            mesh, parent = refine_hexes(mesh.points, mesh.hexes, mask, split_z=bool(seed % 3))
            U = U[parent] + rng.normal(size=(mesh.n_cells, 3)) * 0.1 * float(np.abs(U).max())
            dt = dt * 0.5
        if box:
            from cudaparticlesfoam_amd.cases import box_mesh
            nb = [int(rng.integers(3, 11)) for _ in range(3)]
            if seed % 2:                                 # unit cells, whole or half cells per step along every axis: ties galore
                mesh = box_mesh(*nb)
                cc, _ = mesh.cell_centres_volumes()
                U = rng.choice([-1.0, -0.5, 0.0, 0.5, 1.0], size=(mesh.n_cells, 3)) if seed % 4 == 1 else \
                    np.tile(rng.choice([-1.0, -0.5, 0.5, 1.0], size=3), (mesh.n_cells, 1))
                dt = 1.0
            else:
                lo3 = rng.normal(size=3); ext = rng.uniform(0.3, 2.0, size=3)
                mesh = box_mesh(*nb, lower=tuple(lo3), upper=tuple(lo3 + ext), grading=tuple(rng.choice([0.25, 1.0, 3.0], size=3)))
                U = rng.normal(size=(mesh.n_cells, 3)) * ext * float(rng.choice([0.5, 2.0, 6.0]))
                dt = float(rng.choice([0.05, 0.2]))
            box_refined = seed % 3 == 2
            if box_refined:                              # ... with a random subset of the cells split 2 x 2 x 2: boxes with face groups (LOOKUP 11)
                from cudaparticlesfoam_amd.cases.refine import refine_hexes
                mask = rng.random(mesh.n_cells) < rng.choice([0.1, 0.3])
                mask[int(rng.integers(mesh.n_cells))] = True
                mesh, parent = refine_hexes(mesh.points, mesh.hexes, mask, split_z=True)
                U = U[parent]
        if flat:
            import dataclasses
            from cudaparticlesfoam_amd.cases import box_mesh
            nb = [int(rng.integers(3, 9)), int(rng.integers(3, 8)), int(rng.integers(1, 4))]
            mesh = box_mesh(*nb, upper=(float(rng.uniform(0.5, 3)), float(rng.uniform(0.5, 2)), float(rng.uniform(0.05, 0.5))),
                            grading=tuple(rng.choice([0.3, 1.0, 2.5], size=3)))
            P = np.array(mesh.points, dtype=np.float64)
            sx, sy = rng.uniform(-0.4, 0.4, size=2)
            P[:, 0], P[:, 1] = P[:, 0] + sx * P[:, 1] + 0.05 * P[:, 1] ** 2, P[:, 1] + sy * P[:, 0]      # z stays: straight extrusion
            mesh = dataclasses.replace(mesh, points=P)
            U = rng.normal(size=(mesh.n_cells, 3)) * float(rng.choice([0.5, 2.0, 6.0]))
            U[:, 2] = rng.choice([0.0, -0.0], size=mesh.n_cells)
            dt = float(rng.choice([0.02, 0.1, 0.3]))
        mode = seed % 4
        if mode == 1 and not box and not flat:                                    # axis-aligned flow: whole families of faces have den == 0
            amp = float(np.abs(U).max()) or 1.0
            U = np.zeros_like(U); U[:, seed % 3] = rng.normal(size=U.shape[0]) * amp
        if mode == 2:
            dt = dt * (2 if poly else 4)                 # many cells and walls per step
        t = cw.build(mesh)
        lo, hi = mesh.bounds()
        n = int(rng.integers(1000, 60000))
        if flat:
            n = int(mesh.n_cells * rng.integers(130, 200))
        xyz = rng.uniform(lo - 0.02 * (hi - lo), hi + 0.02 * (hi - lo), size=(n, 3))
        if box and seed % 2:                             # half of the cloud on quarter-cell positions (strictly inside the domain)
            k = n // 2
            xyz[:k] = np.clip(np.round(xyz[:k] * 4.0) / 4.0, lo + 0.25, hi - 0.25)
        ref0 = cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t, nthreads=cw.max_threads)
        x, y, z, c = xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), ref0.copy()
        steps = [1, 5, 24]
        for k in steps:
            cw.step(x, y, z, c, dt, k, t, U, nthreads=cw.max_threads)
        if mixed:
            from test_oracle_mixed import worst_outside
            alive = c >= 0
            w = worst_outside(t, np.stack([x, y, z], 1)[alive], c[alive]) if alive.any() else np.zeros(1)
            # (mode 2 sends particles across the whole block several times per step: more than five reflections, i.e. lost by
            # the reference's own cap -- the plain block loses them too)
            # (poly: steps of up to two unit cells in a grid 1-3 cells thick -- the five-reflection cap takes particles there too)
            if (mode != 2 and not poly and (c[ref0 >= 0] < 0).any()) or w.max() > 1e-9 * float((hi - lo).max()):
                bad += 1
                print("INVARIANT seed %d: %d lost, %d outside their cell (worst %.3e)" %
                      (seed, int((c[ref0 >= 0] < 0).sum()), int((w > 1e-9 * float((hi - lo).max())).sum()), float(w.max())), flush=True)
        for stats in (0, 1):
            for fused in (0, 1):
                ctx = Context(0)
                ctx.set_option("stats", stats)
                if os.environ.get("CPF_FUZZ_VARIANT"):
                    ctx.set_option("step_variant", int(os.environ["CPF_FUZZ_VARIANT"]))
                if flat:
                    ctx.set_option("flat_walk", 0 if (seed + stats + fused) % 5 == 0 else 1)
                elif box and box_refined:
                    ctx.set_option("box_records", (1, 1, 0)[(seed // 2 + stats + 2 * fused) % 3])        # 11 against 3
                elif box:
                    ctx.set_option("stream_lookup", (6, 6, 1, 4)[(seed // 2 + stats + 2 * fused) % 4])
                else:
                    ctx.set_option("stream_lookup", (0, 1) [(seed // 2 + stats) % 2] if mixed else (0, 1, 4)[(seed // 2 + stats) % 3])      # all record-lookup methods of the streaming kernel
                ctx.set_mesh(mesh); ctx.set_velocity(U); ctx.set_particles(xyz)
                ctx.locate_initial()
                _, cell0 = ctx.get_particles()
                ok = np.array_equal(cell0, ref0)
                if (seed + fused) % 2 == 0:
                    ctx.sort_by_cell()
                for k in steps:
                    ctx.step(dt, 0.0, k, L.STEP_FUSE_CYCLES if fused else 0)
                xyzw, cell = ctx.get_particles()
                ok = ok and np.array_equal(cell, c) and np.array_equal(xyzw[:, 0], x) and np.array_equal(xyzw[:, 1], y) \
                    and np.array_equal(xyzw[:, 2], z)
                if flat:                                 # bits: the flat walk's argument is about signs of zeros
                    ok = ok and np.array_equal(xyzw[:, :3].view(np.int64), np.stack([x, y, z], 1).view(np.int64))
                ctx.close()
                if not ok:
                    bad += 1
                    nd = int((cell != c).sum())
                    print("MISMATCH seed %d stats %d fused %d: %d cells differ, max |dx| %.3e" %
                          (seed, stats, fused, nd, float(np.abs(xyzw[:, 0] - x).max())), flush=True)
        if (seed - first) % 10 == 9:
            print("... %d cases, %d mismatches, %.0f s" % (seed - first + 1, bad, time.time() - t0), flush=True)
    print("fuzz done: %d cases x 4 configurations, %d mismatches" % (count, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
