"""TEST INFRASTRUCTURE ONLY.  ctypes front-ends for the three CPU checkers:

* ``RefLib``   -- oracle/_ref/libref_rtxadvect.so: the reference's OWN device functions compiled
                  for CPU by oracle/build_ref.sh (exists only where it was built from /root/reference).
* ``TetWalk``  -- oracle/liboracle_tetwalk.so: plain-C restatement of the reference algorithm.
* ``CellWalk`` -- oracle/liboracle_cellwalk.so: polyhedral-cell formulation (what the kernels implement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_lp = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_up = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> None:
    """Compile the C restatements (and oracle/_ref when the reference tree is present)."""
    need = force or not (os.path.exists(os.path.join(HERE, "liboracle_tetwalk.so"))
                         and os.path.exists(os.path.join(HERE, "liboracle_cellwalk.so")))
    src_newer = False
    for so, src in (("liboracle_tetwalk.so", "tetwalk.c"), ("liboracle_cellwalk.so", "cellwalk.c")):
        a, b = os.path.join(HERE, so), os.path.join(HERE, src)
        if os.path.exists(a) and os.path.getmtime(b) > os.path.getmtime(a):
            src_newer = True
    ref_missing = (not (os.path.exists(os.path.join(HERE, "_ref", "libref_rtxadvect.so"))
                        and os.path.exists(os.path.join(HERE, "_ref", "libref_rtxadvect_fma.so")))
                   and os.path.isdir(os.environ.get("CPF_REFERENCE", "/root/reference")))
    if need or src_newer or ref_missing:
        subprocess.run(["make", "-C", HERE, "-s"] + (["-B"] if force else []), check=True)


def usable_threads(hw_threads: int) -> int:
    """Threads the TESTS use: capped (default 16, env CPF_ORACLE_THREADS).  GPU boxes report hundreds of
    hardware threads that the container may only get a slice of; an oversubscribed OpenMP team crawls."""
    cap = int(os.environ.get("CPF_ORACLE_THREADS", "16"))
    return max(1, min(int(hw_threads), cap))


def have_ref() -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", "libref_rtxadvect.so"))


def have_ref_fma() -> bool:
    """The CONTRACTING build of the same splices (oracle/build_ref.sh: -ffp-contract=fast -mfma, as nvcc fuses by default)."""
    return os.path.exists(os.path.join(HERE, "_ref", "libref_rtxadvect_fma.so"))


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class TetMeshTables:
    """Reference device-mesh arrays (cuda/DeviceTetMesh.cuh:26-37) as flat numpy arrays."""

    def __init__(self, positions, tets, tetvel, facets, tetfacets, faceinfo):
        self.positions = _c(positions, np.float64)
        self.tets = _c(tets, np.int32)
        self.tetvel = None if tetvel is None else _c(tetvel, np.float64)
        self.facets = _c(facets, np.int32)
        self.tetfacets = _c(tetfacets, np.int32)
        self.faceinfo = _c(faceinfo, np.int32)


class _TetApi:
    """Shared driver for the two tet-walk libraries (same flat signatures, different prefixes)."""

    prefix = ""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.hw_threads = getattr(self.lib, self.prefix + "max_threads")()
        self.max_threads = usable_threads(self.hw_threads)

    # ---- face table
    def face_table(self, positions, tets):
        positions = _c(positions, np.float64); tets = _c(tets, np.int32)
        nT = tets.shape[0]; cap = 4 * nT
        facets = np.empty((cap, 4), np.int32); tetfacets = np.empty((nT, 4), np.int32)
        faceinfo = np.empty((cap, 2), np.int32)
        fn = getattr(self.lib, self.prefix + "face_table")
        fn.restype = C.c_int
        fn.argtypes = [_dp, C.c_int, _ip, C.c_int, _ip, _ip, _ip, C.c_int]
        nF = fn(positions, positions.shape[0], tets, nT, facets, tetfacets, faceinfo, cap)
        if nF < 0:
            raise RuntimeError("face table failed (%d)" % nF)
        return facets[:nF].copy(), tetfacets, faceinfo[:nF].copy()

    def tables(self, positions, tets, tetvel=None) -> TetMeshTables:
        f, tf, fi = self.face_table(positions, tets)
        return TetMeshTables(positions, tets, tetvel, f, tf, fi)


class RefLib(_TetApi):
    prefix = "ref_"

    def __init__(self, fma: bool = False):
        """fma=True: the contracting build (libref_rtxadvect_fma.so) -- same functions, multiply-adds fused."""
        super().__init__(os.path.join(HERE, "_ref", "libref_rtxadvect_fma.so" if fma else "libref_rtxadvect.so"))
        L = self.lib
        L.ref_init_particles.argtypes = [_dp, C.c_int, _dp, _dp, C.c_int]
        L.ref_advect.argtypes = [_dp, _ip, _dp, _dp, C.c_double, C.c_int, _ip, _dp, _dp, C.c_int]
        L.ref_advect_vertex.argtypes = [_dp, _ip, _dp, _dp, C.c_double, C.c_int, _ip, _dp, _dp, C.c_int]
        L.ref_locate.argtypes = [_dp, _ip, _dp, C.c_int, _ip, _dp, _ip, _ip, _ip, C.c_int]
        L.ref_reflect.argtypes = [_dp, _ip, _dp, _dp, C.c_int, _ip, _dp, _ip, _ip, _ip, C.c_int]
        L.ref_move.argtypes = [_dp, _dp, _ip, C.c_int, C.c_int]
        L.ref_bary_query.argtypes = [_dp, _ip, C.c_int, _dp, _ip, _ip, _ip, _ip, C.c_int]
        L.ref_bary_query_disp.argtypes = [_dp, _dp, _ip, C.c_int, _dp, _ip, _ip, _ip, _ip, C.c_int]
        L.ref_cycles.argtypes = [_dp, _ip, _dp, _dp, C.c_double, C.c_int, C.c_int, _ip, _dp, _dp, _ip, _ip, _ip, C.c_int]
        L.ref_box_mesh.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _ip, C.c_int, C.c_int]
        L.ref_box_mesh.restype = C.c_int
        L.ref_write_vtu.argtypes = [C.c_uint, _dp, _dp, _ip, C.c_int, _ip]
        L.ref_traj_add.argtypes = [_dp, C.c_int]
        L.ref_traj_save_obj.argtypes = [C.c_char_p]
        L.ref_traj_write_vtk.argtypes = [C.c_char_p]

    def write_vtu(self, directory, ti, particles, vels, tet_ids, convex_ids):
        """The reference's writeParticles2VTU (cuda/utils.cpp:144-283) on host arrays: <directory>/particle_%04d.vtu."""
        cwd = os.getcwd()
        os.chdir(directory)
        try:
            self.lib.ref_write_vtu(int(ti), _c(particles, np.float64), _c(vels, np.float64), _c(tet_ids, np.int32),
                                   int(np.asarray(tet_ids).shape[0]), _c(convex_ids, np.int32))
        finally:
            os.chdir(cwd)
        return os.path.join(directory, "particle_%04d.vtu" % ti)

    def trajectories(self, samples, obj_path, vtk_path):
        """The reference's addToTrajectories / saveTrajectories / writeStreamline2VTK (cuda/utils.cpp:7-94) on host samples
        (a list of [n][4] particle arrays, w = 0: inactive)."""
        self.lib.ref_traj_reset()
        for P in samples:
            P = _c(P, np.float64)
            self.lib.ref_traj_add(P, P.shape[0])
        self.lib.ref_traj_save_obj(obj_path.encode())
        self.lib.ref_traj_write_vtk(vtk_path.encode())

    def box_mesh(self, nx, ny, nz):
        nV = (nx + 1) * (ny + 1) * (nz + 1); nT = 6 * nx * ny * nz
        pos = np.empty((nV, 3)); tets = np.empty((nT, 4), np.int32)
        if self.lib.ref_box_mesh(nx, ny, nz, pos, tets, nV, nT) != nT:
            raise RuntimeError("ref_box_mesh")
        return pos, tets

    def init_particles(self, n, lower, upper, nthreads=1):
        P = np.zeros((n, 4))
        self.lib.ref_init_particles(P, n, _c(lower, np.float64), _c(upper, np.float64), nthreads)
        return P

    def advect(self, P, ids, vels, disps, dt, m: TetMeshTables, nthreads=1):
        self.lib.ref_advect(P, ids, vels, disps, dt, P.shape[0], m.tets, m.positions, m.tetvel, nthreads)

    def advect_vertex(self, P, ids, vels, disps, dt, m: TetMeshTables, vertvel, nthreads=1):
        self.lib.ref_advect_vertex(P, ids, vels, disps, dt, P.shape[0], m.tets, m.positions, _c(vertvel, np.float64), nthreads)

    def locate(self, P, ids, disps, m, nthreads=1):
        self.lib.ref_locate(P, ids, disps, P.shape[0], m.tets, m.positions, m.tetfacets, m.facets, m.faceinfo, nthreads)

    def reflect(self, P, ids, disps, vels, m, nthreads=1):
        self.lib.ref_reflect(P, ids, disps, vels, P.shape[0], m.tets, m.positions, m.tetfacets, m.facets, m.faceinfo, nthreads)

    def move(self, P, disps, ids, nthreads=1):
        self.lib.ref_move(P, disps, ids, P.shape[0], nthreads)

    def bary_query(self, P, ids, m, nthreads=1):
        self.lib.ref_bary_query(P, ids, P.shape[0], m.positions, m.tets, m.facets, m.tetfacets, m.faceinfo, nthreads)

    def bary_query_disp(self, P, disps, ids, m, nthreads=1):
        self.lib.ref_bary_query_disp(P, disps, ids, P.shape[0], m.positions, m.tets, m.facets, m.tetfacets, m.faceinfo, nthreads)

    def cycles(self, P, ids, vels, disps, dt, k, m, nthreads=1):
        self.lib.ref_cycles(P, ids, vels, disps, dt, P.shape[0], k, m.tets, m.positions, m.tetvel, m.tetfacets,
                            m.facets, m.faceinfo, nthreads)


class TetWalk(_TetApi):
    prefix = "orc_"

    def __init__(self):
        super().__init__(os.path.join(HERE, "liboracle_tetwalk.so"))
        L = self.lib
        L.orc_init_particles.argtypes = [_dp, C.c_int, _dp, _dp, C.c_int]
        L.orc_advect.argtypes = [_dp, _ip, _dp, _dp, C.c_double, C.c_int, _ip, _dp, _dp, C.c_int]
        L.orc_advect_vertex.argtypes = [_dp, _ip, _dp, _dp, C.c_double, C.c_int, _ip, _dp, _dp, C.c_int]
        L.orc_locate.argtypes = [_dp, _ip, _dp, C.c_int, _dp, _ip, _ip, _ip, C.c_int]
        L.orc_reflect.argtypes = [_dp, _ip, _dp, _dp, C.c_int, _dp, _ip, _ip, _ip, C.c_int]
        L.orc_move.argtypes = [_dp, _dp, C.c_int, C.c_int]
        L.orc_bary_query.argtypes = [_dp, _ip, C.c_int, _dp, _ip, _ip, _ip, C.c_int]
        L.orc_cycles.argtypes = [_dp, _ip, _dp, _dp, C.c_double, C.c_int, C.c_int, _ip, _dp, _dp, _ip, _ip, _ip, C.c_int]

    def init_particles(self, n, lower, upper, order=1):
        P = np.zeros((n, 4))
        self.lib.orc_init_particles(P, n, _c(lower, np.float64), _c(upper, np.float64), order)
        return P

    def advect(self, P, ids, vels, disps, dt, m, nthreads=1):
        self.lib.orc_advect(P, ids, vels, disps, dt, P.shape[0], m.tets, m.positions, m.tetvel, nthreads)

    def advect_vertex(self, P, ids, vels, disps, dt, m, vertvel, nthreads=1):
        self.lib.orc_advect_vertex(P, ids, vels, disps, dt, P.shape[0], m.tets, m.positions, _c(vertvel, np.float64), nthreads)

    def locate(self, P, ids, disps, m, nthreads=1):
        self.lib.orc_locate(P, ids, disps, P.shape[0], m.positions, m.tetfacets, m.facets, m.faceinfo, nthreads)

    def reflect(self, P, ids, disps, vels, m, nthreads=1):
        self.lib.orc_reflect(P, ids, disps, vels, P.shape[0], m.positions, m.tetfacets, m.facets, m.faceinfo, nthreads)

    def move(self, P, disps, ids=None, nthreads=1):
        self.lib.orc_move(P, disps, P.shape[0], nthreads)

    def bary_query(self, P, ids, m, nthreads=1):
        self.lib.orc_bary_query(P, ids, P.shape[0], m.positions, m.tets, m.tetfacets, m.faceinfo, nthreads)

    def cycles(self, P, ids, vels, disps, dt, k, m, nthreads=1):
        self.lib.orc_cycles(P, ids, vels, disps, dt, P.shape[0], k, m.tets, m.positions, m.tetvel, m.tetfacets,
                            m.facets, m.faceinfo, nthreads)


class CellTables:
    """cw_build's tables: one slot per distinct PLANE of a cell; coplanar internal faces form a face group
    (nbr = GROUP_BASE + g, the pieces' cells in group_nbr[group_off[g]:group_off[g+1]])."""
    GROUP_BASE = -2 ** 31 + 16

    def __init__(self, cell_off, planes, nbr, n_cells, group_off=None, group_nbr=None):
        self.cell_off, self.planes, self.nbr, self.n_cells = cell_off, planes, nbr, n_cells
        self.group_off = np.zeros(1, np.int32) if group_off is None else group_off
        self.group_nbr = np.zeros(1, np.int32) if group_nbr is None else group_nbr

    @property
    def n_groups(self):
        return self.group_off.shape[0] - 1


class CellWalk:
    def __init__(self):
        path = os.path.join(HERE, "liboracle_cellwalk.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = self.lib = C.CDLL(path)
        L.cw_build.restype = C.c_int
        L.cw_build.argtypes = [_dp, C.c_int, _ip, _ip, C.c_int, _ip, _ip, C.c_int, C.c_int, _ip, _dp, _ip, _ip, _ip, _ip]
        L.cw_step.argtypes = [_dp, _dp, _dp, _ip, C.c_void_p, C.c_int, C.c_double, C.c_int, _ip, _dp, _ip, _ip, _ip, _dp,
                              C.c_int, _lp, C.c_double, C.c_void_p, C.c_uint32, C.c_uint32]
        L.cw_locate_initial.argtypes = [_dp, _dp, _dp, _ip, C.c_int, C.c_int, _ip, _dp, C.c_int]
        L.cw_step_count.argtypes = [_dp, _dp, _dp, _ip, C.c_int, C.c_double, _ip, _dp, _ip, _ip, _ip, _dp, C.c_int, _ip, _ip]
        L.cw_philox4x32_10.argtypes = [_up, _up, _up]
        L.cw_philox4x32.argtypes = [_up, _up, C.c_int, _up]
        L.cw_normal3.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, _dp]
        L.cw_normal3_words.argtypes = [_up, _dp]
        L.cw_scan_min_radius_word.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int64, C.POINTER(C.c_int64),
                                              C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.cw_advect_vertex.argtypes = [_dp, _ip, _dp, _dp, C.c_double, C.c_int, _ip, C.c_int, _dp, _dp, C.c_int]
        self.hw_threads = L.cw_max_threads()
        self.max_threads = usable_threads(self.hw_threads)

    def build(self, mesh) -> CellTables:
        ncf = int(mesh.n_faces + mesh.n_internal)
        off = np.empty(mesh.n_cells + 1, np.int32); planes = np.empty((ncf, 4)); nbr = np.empty(ncf, np.int32)
        goff = np.zeros(ncf + 1, np.int32); gnbr = np.zeros(ncf, np.int32); ng = np.zeros(1, np.int32)
        r = self.lib.cw_build(_c(mesh.points, np.float64), mesh.n_points, _c(mesh.face_offsets, np.int32),
                              _c(mesh.face_verts, np.int32), mesh.n_faces, _c(mesh.owner, np.int32),
                              _c(mesh.neighbour, np.int32), mesh.n_internal, mesh.n_cells, off, planes, nbr, goff, gnbr, ng)
        if r < 0 or r != off[-1]:
            raise RuntimeError("cw_build")
        g = int(ng[0])
        return CellTables(off, planes[:r].copy(), nbr[:r].copy(), mesh.n_cells, goff[:g + 1].copy(),
                          gnbr[:max(int(goff[g]), 1)].copy())

    def advect_vertex(self, P, cells, vels, disps, dt, tets, tets_per_cell, positions, vertvel, nthreads=1):
        """cpf_stage_advect_vertex's CPU statement: "VertexVelocity" advect on cell ids (tets: [nCells*tpc][4])."""
        self.lib.cw_advect_vertex(P, _c(cells, np.int32), vels, disps, dt, P.shape[0], _c(tets, np.int32), tets_per_cell,
                                  _c(positions, np.float64), _c(vertvel, np.float64), nthreads)

    def step(self, x, y, z, cell, dt, cycles, t: CellTables, U, vel_out: Optional[np.ndarray] = None,
             nthreads=1, D=0.0, gid: Optional[np.ndarray] = None, step0=0, seed=0):
        stats = np.zeros(3, np.int64)
        U = _c(U, np.float64)
        vp = None if vel_out is None else vel_out.ctypes.data_as(C.c_void_p)
        gp = None if gid is None else _c(gid, np.int64).ctypes.data_as(C.c_void_p)
        self.lib.cw_step(x, y, z, cell, vp, x.shape[0], dt, cycles, t.cell_off, t.planes, t.nbr, t.group_off, t.group_nbr,
                         U, nthreads, stats, D, gp, step0, seed)
        return stats

    def step_count(self, x, y, z, cell, dt, t: CellTables, U, nthreads=1):
        """One cycle (D = 0) in place; returns per-particle (cells visited, wall reflections)."""
        n = x.shape[0]
        visits = np.zeros(n, np.int32); refl = np.zeros(n, np.int32)
        self.lib.cw_step_count(x, y, z, cell, n, dt, t.cell_off, t.planes, t.nbr, t.group_off, t.group_nbr, _c(U, np.float64),
                               nthreads, visits, refl)
        return visits, refl

    def locate_initial(self, x, y, z, t: CellTables, nthreads=1):
        cell = np.empty(x.shape[0], np.int32)
        self.lib.cw_locate_initial(_c(x, np.float64), _c(y, np.float64), _c(z, np.float64), cell, x.shape[0],
                                   t.n_cells, t.cell_off, t.planes, nthreads)
        return cell

    def philox(self, ctr, key, rounds=10):
        """Philox4x32-`rounds` (10: Random123's default, the round count of its known-answer vectors; the kernels draw at 7)"""
        out = np.zeros(4, np.uint32)
        self.lib.cw_philox4x32(_c(ctr, np.uint32), _c(key, np.uint32), int(rounds), out)
        return out

    def normal3(self, gid, step, seed):
        out = np.zeros(3)
        self.lib.cw_normal3(int(gid), int(step), int(seed), out)
        return out

    def normal3_words(self, words):
        """the deviate transform alone, on four given 32-bit words"""
        out = np.zeros(3)
        self.lib.cw_normal3_words(_c(np.asarray(words), np.uint32), out)
        return out

    def scan_min_radius_word(self, seed, step0, n_steps, n):
        """(gid, step, word): where in [0, n) x [step0, step0 + n_steps) the first radius word is smallest"""
        g, st, w = C.c_int64(0), C.c_uint32(0), C.c_uint32(0)
        self.lib.cw_scan_min_radius_word(int(seed), int(step0), int(n_steps), int(n), C.byref(g), C.byref(st), C.byref(w))
        return int(g.value), int(st.value), int(w.value)
