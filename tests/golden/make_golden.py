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
Usage: python tests/golden/make_golden.py   (from the repo root, in the container with /root/reference)
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
    main()
