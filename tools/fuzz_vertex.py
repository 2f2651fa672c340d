#!/usr/bin/env python3
"""Developer tool (GPU box): randomised differential campaign for the "VertexVelocity" cycle -- the streaming kernel with the cone
locate (step_kernel_stream_vertex; one tet per particle where that is provably the all-tets result) against the rule itself, the
evaluation of all twelve tets on the generic walk (option vertex_fast 0, step_variant 0), bit for bit.  Random graded / sheared hex
blocks (tests/test_oracle_random._case), random vertex velocities, a cloud of which a third sits exactly on or a rounding off the
fans' structure (apexes, corners, edges, internal planes), 6 cycles with reflections, sorted and unsorted clouds, single and fused
launches, with and without the Brownian kick.  python tools/fuzz_vertex.py [first_seed] [count]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    import torch  # noqa: F401  (its HIP runtime first)
    from cudaparticlesfoam_amd import _lib as L
    from cudaparticlesfoam_amd.api import Context
    from test_oracle_random import _case
    bad = streamed = 0
    alive = total = moved = 0
    t0 = time.time()
    fast, full = Context(0), Context(0)
    full.set_option("vertex_fast", 0); full.set_option("step_variant", 0)
    for seed in range(first, first + count):
        rng, mesh, _, _ = _case(seed)
        centres, _ = mesh.cell_centres_volumes()
        pos, tets = mesh.tet_decomposition(centres)
        vU = rng.normal(size=pos.shape) * rng.uniform(0.2, 2.0)
        lo, hi = mesh.bounds()
        n = int(rng.integers(2000, 20000))
        P = rng.uniform(lo, hi, size=(n, 3))
        # a third of the cloud on the fans' structure: mixtures of a random tet's vertices with weights that are often exactly 0
        t = tets[rng.integers(0, tets.shape[0], n // 3)]
        w = rng.dirichlet([0.3, 0.3, 0.3, 0.3], n // 3) * (rng.random((n // 3, 4)) > 0.35)
        w[w.sum(1) == 0, 0] = 1.0
        w /= w.sum(1, keepdims=True)
        P[: n // 3] = np.einsum("nk,nkd->nd", w, pos[t]) + rng.normal(size=(n // 3, 3)) * rng.choice([0.0, 1e-15, 1e-12, 1e-9], size=(n // 3, 1))
        dt = float(rng.uniform(0.05, 0.8) * (hi - lo).min() / max(mesh.n_cells ** (1 / 3), 1) / max(1e-9, np.abs(vU).max()))
        D = float(rng.choice([0.0, 0.0, 1e-3 * (hi - lo).min() ** 2 / dt / 100]))
        fused = bool(rng.integers(0, 2)); do_sort = bool(rng.integers(0, 2))
        out = []
        for ctx in (fast, full):
            ctx.set_mesh(mesh); ctx.set_velocity(np.zeros((mesh.n_cells, 3)))
            ctx.set_tets(pos, tets, 12); ctx.set_vertex_velocity(vU)
            ctx.set_particles(P); ctx.locate_initial()
            if do_sort:
                ctx.sort_by_cell()
            fl = L.STEP_VERTEX_VELOCITY | (L.STEP_FUSE_CYCLES if fused else 0)
            ctx.step(dt, D, 6, fl)
            out.append(ctx.get_particles())
        inside = out[1][1] >= 0
        alive += int(inside.sum()); total += n
        moved += int((np.abs(out[1][0][:, :3] - P).max(1) > 1e-6 * (hi - lo).min())[inside].sum())
        name = fast.step_kernel_name(D, L.STEP_VERTEX_VELOCITY)
        streamed += "stream_vertex" in name
        ok = np.array_equal(out[0][0], out[1][0], equal_nan=True) and np.array_equal(out[0][1], out[1][1])
        if not ok:
            bad += 1
            d = np.nonzero((out[0][0] != out[1][0]).any(1) | (out[0][1] != out[1][1]))[0]
            print("MISMATCH seed %d: %d particles, e.g. %d: %r vs %r (%s)" % (seed, d.size, d[0], out[0][0][d[0]], out[1][0][d[0]], name), flush=True)
        if (seed - first + 1) % 200 == 0:
            print("... %d cases, %d mismatches, %d streamed, %.0f s" % (seed - first + 1, bad, streamed, time.time() - t0), flush=True)
    fast.close(); full.close()
    print("vertex fuzz done: %d cases (%d on step_kernel_stream_vertex), %d mismatches; %d particles, %d still in the mesh after 6 cycles, %d of them moved"
          % (count, streamed, bad, total, alive, moved))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
