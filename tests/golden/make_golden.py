#!/usr/bin/env python3
"""Generates the committed golden vectors by running the REFERENCE's own functions
(oracle/_ref, built by oracle/build_ref.sh from /root/reference -- only possible in the build
container) on seeded inputs.  Only inputs and expected outputs are stored (tests/golden/*.npz).

Cases (reference order of one cycle: advect -> locate -> reflect -> move, src/advect.H:96-161,
ConvexPoly build, diffusionCoeff 0):
  pitz_uniform / pitz_analytic : pitzDaily 12-tets-per-cell decomposition, 512 particles seeded in the
                                 inlet box, dt 1e-4, checkpoints after 1, 10, 100, 1000 cycles
  box_random                   : 10x9x8 hex box, random cell-constant U, 600 particles, dt 0.3,
                                 checkpoints after 1, 20, 100 cycles (heavy wall reflection)
  stages_box                   : every intermediate array of ONE cycle (after advect, locate, reflect, move)
  vertex_box                   : "VertexVelocity" mode (particleAdvectKernel, cuda/particles.cu:244-313) on the same box:
                                 random velocities on the tet-mesh vertices (points ++ cell centres), one cycle's
                                 advect arrays and checkpoints after 1, 20, 60 cycles of advect -> locate -> reflect -> move
  face_table_box               : getBoundaryMesh tables of createBoxMesh(3,2,2)
  init_particles               : cudaInitParticles LCG<16> stream (g++ argument evaluation order)
  *_fma                        : `python tests/golden/make_golden.py fma` -- pitz_uniform / pitz_analytic / box_random once
                                 more (2048 / 2048 / 1024 particles, same seeds and checkpoints; positions and tet ids only)
                                 from the CONTRACTING build of the same reference functions (oracle/_ref/
                                 libref_rtxadvect_fma.so: -ffp-contract=fast -mfma, the way nvcc's default --fmad=true
                                 fuses), plus profiles/r06_fma_sensitivity.json: strict vs contracting build on 400 000
                                 particles (cells that differ, max and 99.99-percentile |dx|/L per checkpoint).  Does NOT
                                 rewrite the files above.
Usage: python tests/golden/make_golden.py [fma]   (from the repo root, in the container with /root/reference)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O                                    # noqa: E402
from oracle.tetmesh import poly_to_tets                            # noqa: E402
from cudaparticlesfoam_amd.cases import box_mesh, pitzdaily as pz  # noqa: E402


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def run_case(ref, mesh, centres, U, xyz, cell0, dt, checkpoints):
    pos, tets, tcell, tu = poly_to_tets(mesh, centres, U)
    m = ref.tables(pos, tets, tu)
    n = xyz.shape[0]
    P = np.zeros((n, 4)); P[:, :3] = xyz; P[:, 3] = 1
    ids = (cell0 * 12).astype(np.int32)
    ref.bary_query(P, ids, m)
    ids0 = ids.copy()
    vels = np.zeros((n, 4)); disps = np.zeros((n, 4))
    out = dict(xyz0=xyz, tet0=ids0, dt=np.float64(dt), checkpoints=np.asarray(checkpoints),
               inputs_sha256=np.array(digest(mesh.points, mesh.face_verts, mesh.owner, mesh.neighbour, U)))
    done = 0
    for k in checkpoints:
        ref.cycles(P, ids, vels, disps, dt, k - done, m, nthreads=ref.max_threads)
        done = k
        out["P_%d" % k] = P.copy(); out["tet_%d" % k] = ids.copy(); out["vel_%d" % k] = vels.copy()
    return out


FMA_CASES = (("pitz_uniform", 2048, 1e-4, (1, 10, 100, 1000)), ("pitz_analytic", 2048, 1e-4, (1, 10, 100, 1000)),
             ("box_random", 1024, 0.3, (1, 20, 100)))


def fma_case_inputs(name, n, cw):
    """(mesh, centres, U, xyz, cell0, L) of an fma case: the seeds of the strict goldens, n particles."""
    if name.startswith("pitz"):
        mesh = pz.pitzdaily_mesh()
        centres, _ = mesh.cell_centres_volumes()
        U = pz.uniform_u(mesh) if name == "pitz_uniform" else pz.analytic_step_u(mesh, centres)
        xyz = pz.uniform_points(12345, n, *pz.INLET_BOX)
    else:
        mesh = box_mesh(10, 9, 8)
        centres, _ = mesh.cell_centres_volumes()
        rng = np.random.default_rng(11)
        U = rng.normal(size=(mesh.n_cells, 3))
        xyz = rng.uniform([0, 0, 0], [10, 9, 8], size=(n, 3))
    t = cw.build(mesh)
    cell0 = cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t, nthreads=cw.max_threads)
    assert (cell0 >= 0).all()
    lo, hi = mesh.bounds()
    return mesh, centres, U, xyz, cell0, float(np.linalg.norm(hi - lo))


def compare_builds(a, b, L, checkpoints):
    """Per checkpoint: particles whose tet / cell differs between two runs, max and 99.99-percentile |dx|/L, count beyond 1e-5."""
    rows = []
    for k in checkpoints:
        d = np.sqrt(((a["P_%d" % k][:, :3] - b["P_%d" % k][:, :3]) ** 2).sum(1)) / L
        ta, tb = a["tet_%d" % k], b["tet_%d" % k]
        rows.append(dict(cycles=int(k), particles=int(d.size), tets_differ=int((ta != tb).sum()),
                         cells_differ=int((ta // 12 != tb // 12).sum()), max_rel=float(d.max()),
                         p9999_rel=float(np.quantile(d, 0.9999)), beyond_1e5=int((d > 1e-5).sum())))
    return rows


def main_fma(n_sensitivity=400000):
    import json
    O.build()
    if not (O.have_ref() and O.have_ref_fma()):
        sys.exit("oracle/_ref (strict and contracting builds) is not built (needs /root/reference)")
    strict, fma, cw = O.RefLib(), O.RefLib(fma=True), O.CellWalk()
    report = {"what": "the reference's own functions (oracle/build_ref.sh splices) built strict (-ffp-contract=off) against the "
                      "same splices built contracting (-ffp-contract=fast -mfma, as nvcc's default --fmad=true fuses): same "
                      "seeded inputs, ConvexPoly cycle advect -> locate -> reflect -> move, D = 0",
              "rel": "|dx| / L, L = bounding-box diagonal of the mesh", "cases": {}}
    for name, n, dt, cps in FMA_CASES:
        mesh, centres, U, xyz, cell0, L = fma_case_inputs(name, n, cw)
        g = run_case(fma, mesh, centres, U, xyz, cell0, dt, cps)
        keep = {k: v for k, v in g.items() if not k.startswith("vel_")}
        for k in cps:
            keep["P_%d" % k] = np.ascontiguousarray(g["P_%d" % k][:, :3])       # (w stays 1: every boundary reflects)
        if name == "box_random":
            keep["U"] = U
        np.savez_compressed(os.path.join(HERE, name + "_fma.npz"), **keep)
        mesh, centres, U, xyz, cell0, L = fma_case_inputs(name, n_sensitivity, cw)
        a = run_case(strict, mesh, centres, U, xyz, cell0, dt, cps)
        b = run_case(fma, mesh, centres, U, xyz, cell0, dt, cps)
        report["cases"][name] = compare_builds(a, b, L, cps)
        print(name, report["cases"][name][-1])
    with open(os.path.join(ROOT, "profiles", "r06_fma_sensitivity.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("fma goldens written to", HERE)


def main():
    O.build()
    if not O.have_ref():
        sys.exit("oracle/_ref is not built (needs /root/reference): cannot generate goldens here")
    ref = O.RefLib()
    cw = O.CellWalk()

    mesh = pz.pitzdaily_mesh()
    centres, _ = mesh.cell_centres_volumes()
    t = cw.build(mesh)
    xyz = pz.uniform_points(12345, 512, *pz.INLET_BOX)
    cell0 = cw.locate_initial(xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), t, nthreads=cw.max_threads)
    assert (cell0 >= 0).all()
    for name, U in (("pitz_uniform", pz.uniform_u(mesh)), ("pitz_analytic", pz.analytic_step_u(mesh, centres))):
        np.savez_compressed(os.path.join(HERE, name + ".npz"),
                            **run_case(ref, mesh, centres, U, xyz, cell0, 1e-4, (1, 10, 100, 1000)))

    bm = box_mesh(10, 9, 8)
    bc, _ = bm.cell_centres_volumes()
    rng = np.random.default_rng(11)
    Ub = rng.normal(size=(bm.n_cells, 3))
    xb = rng.uniform([0, 0, 0], [10, 9, 8], size=(600, 3))
    tb = cw.build(bm)
    cb = cw.locate_initial(xb[:, 0].copy(), xb[:, 1].copy(), xb[:, 2].copy(), tb, nthreads=cw.max_threads)
    case = run_case(ref, bm, bc, Ub, xb, cb, 0.3, (1, 20, 100))
    case["U"] = Ub
    np.savez_compressed(os.path.join(HERE, "box_random.npz"), **case)

    # one cycle, stage by stage (the reference's five wrappers minus Brownian)
    pos, tets, tcell, tu = poly_to_tets(bm, bc, Ub)
    m = ref.tables(pos, tets, tu)
    n = xb.shape[0]
    P = np.zeros((n, 4)); P[:, :3] = xb; P[:, 3] = 1
    ids = (cb * 12).astype(np.int32); ref.bary_query(P, ids, m)
    vels = np.zeros((n, 4)); disps = np.zeros((n, 4))
    st = dict(U=Ub, xyz0=xb, tet0=ids.copy(), dt=np.float64(0.3))
    ref.advect(P, ids, vels, disps, 0.3, m); st.update(adv_P=P.copy(), adv_vel=vels.copy(), adv_disp=disps.copy())
    ref.locate(P, ids, disps, m); st.update(loc_tet=ids.copy())
    ref.reflect(P, ids, disps, vels, m); st.update(ref_P=P.copy(), ref_tet=ids.copy(), ref_disp=disps.copy(), ref_vel=vels.copy())
    ref.move(P, disps, ids); st.update(mov_P=P.copy(), mov_disp=disps.copy())
    np.savez_compressed(os.path.join(HERE, "stages_box.npz"), **st)

    # "VertexVelocity": barycentric interpolation of vertex velocities in the particle's tet
    vv = np.random.default_rng(21).normal(size=(pos.shape[0], 3))
    P = np.zeros((n, 4)); P[:, :3] = xb; P[:, 3] = 1
    ids = (cb * 12).astype(np.int32); ref.bary_query(P, ids, m)
    vels = np.zeros((n, 4)); disps = np.zeros((n, 4))
    vx = dict(vertex_U=vv, xyz0=xb, tet0=ids.copy(), dt=np.float64(0.2), checkpoints=np.asarray((1, 20, 60)),
              inputs_sha256=np.array(digest(bm.points, bm.face_verts, bm.owner, bm.neighbour, vv)))
    done = 0
    for k in (1, 20, 60):
        for _ in range(k - done):
            ref.advect_vertex(P, ids, vels, disps, 0.2, m, vv)
            if done == 0 and _ == 0:
                vx.update(adv_vel=vels.copy(), adv_disp=disps.copy())
            ref.locate(P, ids, disps, m); ref.reflect(P, ids, disps, vels, m); ref.move(P, disps, ids)
        done = k
        vx["P_%d" % k] = P.copy(); vx["tet_%d" % k] = ids.copy(); vx["vel_%d" % k] = vels.copy()
    np.savez_compressed(os.path.join(HERE, "vertex_box.npz"), **vx)

    bp, bt = ref.box_mesh(3, 2, 2)
    f, tf, fi = ref.face_table(bp, bt)
    np.savez_compressed(os.path.join(HERE, "face_table_box.npz"), positions=bp, tets=bt, facets=f, tetfacets=tf,
                        faceinfo=fi)
    lo, hi = np.array(pz.PARTICLE_DICT["seedingBox"][0]), np.array(pz.PARTICLE_DICT["seedingBox"][1])
    np.savez_compressed(os.path.join(HERE, "init_particles.npz"), lower=lo, upper=hi,
                        P=ref.init_particles(1000, lo, hi))
    print("goldens written to", HERE)


if __name__ == "__main__":
    main_fma() if sys.argv[1:] == ["fma"] else main()
