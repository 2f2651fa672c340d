"""Python host side above the C-ABI.

``Context`` is a thin object wrapper over ``include/cpf.h`` (numpy in / numpy out).
``CudaParticles`` mirrors the reference's two header fragments -- the only "operator
interface" the reference has for this path:

* ``CudaParticles(mesh, U, dict)``        == ``#include "initCuda.H"`` (src/initCuda.H:33-205)
* ``CudaParticles.advect(time, deltaT)``  == ``#include "advect.H"``   (src/advect.H:33-205)

with the same dictionary keys and defaults (src/initCuda.H:50-57), the same
``nCycles = max(ceil(deltaT/dt), 1)`` sub-cycling (src/advect.H:36-37) and the same
output cadence (src/advect.H:166).  All compute happens in the HIP library.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Dict, Optional

import numpy as np

from . import _lib as L


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One GPU, one mesh, one (optional) context-owned particle cloud."""

    def __init__(self, device: int = 0):
        self.lib = L.load()
        h = C.c_void_p()
        st = self.lib.cpf_create(device, C.byref(h))
        if st != L.CPF_OK:
            raise L.CpfError(st, (self.lib.cpf_last_error(None) or b"").decode())
        self.h = h
        self.device = device
        self.n_cells = 0

    # -- plumbing
    def _ck(self, st):
        L.check(self.lib, self.h, st)

    def close(self):
        for sh in list(getattr(self, "_shards", [])):        # shards made on this context (parallel.ShardedCloud) borrow it
            sh.close()
        if getattr(self, "h", None):
            self.lib.cpf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, hip_stream: int):
        self._ck(self.lib.cpf_set_stream(self.h, C.c_void_p(hip_stream)))

    def use_own_stream(self):
        self._ck(self.lib.cpf_use_own_stream(self.h))

    def synchronize(self):
        self._ck(self.lib.cpf_synchronize(self.h))

    # -- mesh / velocity
    def set_mesh(self, mesh):
        """``mesh`` exposes points, face_offsets, face_verts, owner, neighbour, n_cells (cases.PolyMesh)."""
        lab = np.int64 if np.asarray(mesh.owner).dtype == np.int64 else np.int32
        pts = np.ascontiguousarray(mesh.points, dtype=np.float64)
        fo = np.ascontiguousarray(mesh.face_offsets, dtype=lab)
        fv = np.ascontiguousarray(mesh.face_verts, dtype=lab)
        ow = np.ascontiguousarray(mesh.owner, dtype=lab)
        ne = np.ascontiguousarray(mesh.neighbour, dtype=lab)
        fn = self.lib.cpf_set_mesh_l64 if lab == np.int64 else self.lib.cpf_set_mesh
        self._ck(fn(self.h, _ptr(pts), pts.shape[0], _ptr(fo), _ptr(fv), ow.shape[0], _ptr(ow), _ptr(ne), ne.shape[0],
                    int(mesh.n_cells)))
        self.n_cells = int(mesh.n_cells)

    def set_mesh_parts(self, parts):
        """Rank-direct ingest: ``parts`` = the pieces of a decomposed mesh in rank order (each exposes the PolyMesh
        fields); they are stitched on the host (cpf_merge_mesh_parts) and uploaded as one mesh."""
        arr, keep = pack_mesh_parts(parts)
        r = self.lib.cpf_set_mesh_parts(self.h, arr, len(parts))
        if r == L.CPF_ERR_MESH and not self.lib.cpf_last_error(self.h):
            raise L.CpfError(r, (self.lib.cpf_merge_last_error() or b"").decode())
        self._ck(r)
        self.n_cells = int(sum(int(p.n_cells) for p in parts))

    def mesh_info(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.lib.cpf_mesh_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(n_cells=a.value, n_slots=b.value, device_bytes=c.value)

    def mesh_tables(self):
        info = self.mesh_info()
        off = np.empty(info["n_cells"] + 1, np.int32)
        planes = np.empty((info["n_slots"], 4), np.float64)
        nbr = np.empty(info["n_slots"], np.int32)
        self._ck(self.lib.cpf_get_mesh_tables(self.h, _ptr(off), _ptr(planes), _ptr(nbr)))
        return off, planes, nbr

    def mesh_flags(self) -> Dict[str, int]:
        """Which shortcuts of the walk the mesh layer found live: all_hex, z_layered, z_thin, mixed (include/cpf.h)."""
        v = [C.c_int32(0) for _ in range(4)]
        self._ck(self.lib.cpf_get_mesh_flags(self.h, *[C.byref(x) for x in v]))
        return dict(all_hex=v[0].value, z_layered=v[1].value, z_thin=v[2].value, mixed=v[3].value)

    def mesh_groups(self):
        """Face groups of the mesh (coplanar faces of a cell that lead to different cells share one slot): (group_off
        [n_groups + 1], group_nbr); slot neighbour id INT32_MIN + 16 + g marks group g."""
        ng, nm = C.c_int64(0), C.c_int64(0)
        self._ck(self.lib.cpf_get_mesh_groups(self.h, C.byref(ng), C.byref(nm), None, None))
        off = np.zeros(ng.value + 1, np.int32); nbr = np.zeros(max(nm.value, 1), np.int32)
        self._ck(self.lib.cpf_get_mesh_groups(self.h, None, None, _ptr(off), _ptr(nbr)))
        return off, nbr[:nm.value]

    def set_velocity(self, U):
        U = np.ascontiguousarray(U, dtype=np.float64)
        if U.ndim != 2 or U.shape[1] != 3:
            raise ValueError("U must be (nCells, 3)")
        self._ck(self.lib.cpf_set_velocity(self.h, _ptr(U), U.shape[0]))

    def set_tets(self, positions, tets, tets_per_cell: int = 12):
        """The tet decomposition the "VertexVelocity" advect mode needs (src/initCuda.H:86-124): positions =
        mesh.points() ++ mesh.C(), tets [nCells * tets_per_cell][4] in cell order."""
        pos = np.ascontiguousarray(positions, dtype=np.float64); t = np.ascontiguousarray(tets, dtype=np.int32)
        self._ck(self.lib.cpf_set_tets(self.h, _ptr(pos), pos.shape[0], _ptr(t), t.shape[0], int(tets_per_cell)))

    def set_vertex_velocity(self, vertex_u):
        u = np.ascontiguousarray(vertex_u, dtype=np.float64)
        self._ck(self.lib.cpf_set_vertex_velocity(self.h, _ptr(u), u.shape[0]))

    def set_velocity_dev(self, ptr: int, n_cells: int):
        self._ck(self.lib.cpf_set_velocity_dev(self.h, C.c_void_p(ptr), n_cells))

    # -- context-owned cloud
    def seed_box(self, n: int, lower, upper, order: int = 1):
        lo = np.ascontiguousarray(lower, dtype=np.float64); hi = np.ascontiguousarray(upper, dtype=np.float64)
        self._ck(self.lib.cpf_seed_box(self.h, int(n), _ptr(lo), _ptr(hi), order))

    def set_particles(self, xyz, cell=None):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        c = None if cell is None else np.ascontiguousarray(cell, dtype=np.int32)
        self._ck(self.lib.cpf_set_particles(self.h, xyz.shape[0], _ptr(xyz), _ptr(c)))

    def locate_initial(self) -> int:
        out = C.c_int64()
        self._ck(self.lib.cpf_locate_initial(self.h, C.byref(out)))
        return out.value

    def step(self, dt: float, D: float = 0.0, n_cycles: int = 1, flags: int = 0):
        self._ck(self.lib.cpf_step(self.h, dt, D, n_cycles, flags))

    def sort_by_cell(self):
        self._ck(self.lib.cpf_sort_by_cell(self.h))

    @property
    def n(self) -> int:
        out = C.c_int64()
        self._ck(self.lib.cpf_num_particles(self.h, C.byref(out)))
        return out.value

    def get_particles(self, want_vel: bool = False):
        n = self.n
        xyzw = np.empty((n, 4)); cell = np.empty(n, np.int32)
        vel = np.empty((n, 4)) if want_vel else None
        self._ck(self.lib.cpf_get_particles(self.h, _ptr(xyzw), _ptr(cell), _ptr(vel)))
        return (xyzw, cell, vel) if want_vel else (xyzw, cell)

    def counters(self) -> Dict[str, int]:
        out = np.zeros(4, np.int64)
        self._ck(self.lib.cpf_get_counters(self.h, _ptr(out)))
        return dict(particle_steps=int(out[0]), cells_visited=int(out[1]), reflections=int(out[2]), lost=int(out[3]))

    def set_option(self, key: str, value: float):
        self._ck(self.lib.cpf_set_option(self.h, key.encode(), float(value)))

    def step_kernel_name(self, D: float = 0.0, flags: int = 0) -> str:
        buf = C.create_string_buffer(200)
        self._ck(self.lib.cpf_step_kernel_name(self.h, float(D), int(flags), buf, 200))
        return buf.value.decode()

    def set_seed(self, seed: int):
        self._ck(self.lib.cpf_set_seed(self.h, seed & 0xFFFFFFFF))

    # -- device-array level (pointers as ints, e.g. torch.Tensor.data_ptr())
    def step_dev(self, x, y, z, cell, gid, vel, n, dt, D=0.0, step0=0, n_cycles=1, flags=0):
        self._ck(self.lib.cpf_step_dev(self.h, x, y, z, cell, gid, vel, n, dt, D, step0, n_cycles, flags))

    def locate_initial_dev(self, x, y, z, cell, n):
        self._ck(self.lib.cpf_locate_initial_dev(self.h, x, y, z, cell, n))

    def seed_box_dev(self, x, y, z, first, n, lower, upper, order=1):
        lo = np.ascontiguousarray(lower, dtype=np.float64); hi = np.ascontiguousarray(upper, dtype=np.float64)
        self._ck(self.lib.cpf_seed_box_dev(self.h, x, y, z, first, n, _ptr(lo), _ptr(hi), order))

    def sort_by_cell_dev(self, x, y, z, cell, gid, n):
        self._ck(self.lib.cpf_sort_by_cell_dev(self.h, x, y, z, cell, gid, n))

    def sort_by_cell_dev_to(self, x, y, z, cell, gid, ox, oy, oz, ocell, ogid, n):
        self._ck(self.lib.cpf_sort_by_cell_dev_to(self.h, x, y, z, cell, gid, ox, oy, oz, ocell, ogid, n))

    def pack_leavers_dev(self, x, y, z, cell, gid, n, cell_lo, n_ranks, my_rank, sendbuf, send_cap, counts, n_stay):
        self._ck(self.lib.cpf_pack_leavers_dev(self.h, x, y, z, cell, gid, n, cell_lo, n_ranks, my_rank, sendbuf,
                                               send_cap, counts, n_stay))

    def cell_histogram_dev(self, cell, n, scale, weights):
        self._ck(self.lib.cpf_cell_histogram_dev(self.h, cell, n, scale, weights))

    def cell_ranges_dev(self, weights, n_ranks, cell_lo):
        self._ck(self.lib.cpf_cell_ranges_dev(self.h, weights, n_ranks, cell_lo))

    def unpack_arrivals_dev(self, x, y, z, cell, gid, n_stay, recvbuf, n_recv):
        self._ck(self.lib.cpf_unpack_arrivals_dev(self.h, x, y, z, cell, gid, n_stay, recvbuf, n_recv))

    def write_vtu(self, path: str) -> float:
        """particle_%04d.vtu in the reference's layout, synchronously; returns the total kinetic energy."""
        ke = C.c_double()
        r = self.lib.cpf_write_vtu(self.h, str(path).encode(), C.byref(ke))
        if r not in (L.CPF_OK, L.CPF_WARN_NAN):
            self._ck(r)
        return ke.value

    def write_vtu_async(self, path: str, want_ke: bool = True):
        """Same frame behind the caller's back: a device-side snapshot (one kernel), then copy, energy sum, formatting and file
        I/O on a worker thread (one frame in flight per context).  want_ke=False: returns None at once -- the step loop is not
        held up by PCIe; True: waits for the copy and returns the total kinetic energy."""
        ke = C.c_double()
        r = self.lib.cpf_write_vtu_async(self.h, str(path).encode(), C.byref(ke) if want_ke else None)
        if r not in (L.CPF_OK, L.CPF_WARN_NAN):
            self._ck(r)
        return ke.value if want_ke else None

    def write_vtu_wait(self):
        r = self.lib.cpf_write_vtu_wait(self.h)
        if r not in (L.CPF_OK, L.CPF_WARN_NAN):
            self._ck(r)

    def timing_enable(self, on: bool = True):
        self._ck(self.lib.cpf_timing_enable(self.h, int(on)))

    def timing_read(self):
        a, b = C.c_int64(), C.c_double()
        self._ck(self.lib.cpf_timing_read(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def timing_poll(self):
        """Like timing_read but never waits: only launches that already finished are drained."""
        a, b = C.c_int64(), C.c_double()
        self._ck(self.lib.cpf_timing_poll(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value


def build_mesh_tables_host(mesh):
    """What ``Context.set_mesh(mesh)`` would build and upload, computed on the host alone (no GPU, no context): a dict with
    cell_off, planes (n_slots, 4), nbr, group_off, group_nbr -- one slot per distinct plane of a cell, face groups for the
    coplanar pieces of a split face (``include/cpf.h``: cpf_build_mesh_tables_host)."""
    lib = L.load()
    a = [np.ascontiguousarray(mesh.points, dtype=np.float64), np.ascontiguousarray(mesh.face_offsets, dtype=np.int32),
         np.ascontiguousarray(mesh.face_verts, dtype=np.int32), np.ascontiguousarray(mesh.owner, dtype=np.int32),
         np.ascontiguousarray(mesh.neighbour, dtype=np.int32)]
    ns, ng, nm = C.c_int64(0), C.c_int64(0), C.c_int64(0)

    def call(*out):
        st = lib.cpf_build_mesh_tables_host(_ptr(a[0]), mesh.n_points, _ptr(a[1]), _ptr(a[2]), mesh.n_faces, _ptr(a[3]), _ptr(a[4]),
                                            mesh.n_internal, mesh.n_cells, C.byref(ns), C.byref(ng), C.byref(nm), *out)
        if st != L.CPF_OK:
            raise L.CpfError(st, "cpf_build_mesh_tables_host")
    call(None, None, None, None, None)
    off = np.empty(mesh.n_cells + 1, np.int32); planes = np.empty((ns.value, 4), np.float64); nbr = np.empty(ns.value, np.int32)
    goff = np.empty(ng.value + 1, np.int32); gnbr = np.empty(max(nm.value, 1), np.int32)
    call(_ptr(off), _ptr(planes), _ptr(nbr), _ptr(goff), _ptr(gnbr))
    return dict(cell_off=off, planes=planes, nbr=nbr, group_off=goff, group_nbr=gnbr[:nm.value])


def mesh_flags_host(mesh):
    """What ``Context.mesh_flags()`` would report after ``set_mesh(mesh)``, computed on the host alone (cpf_mesh_flags_host)."""
    lib = L.load()
    a = [np.ascontiguousarray(mesh.points, dtype=np.float64), np.ascontiguousarray(mesh.face_offsets, dtype=np.int32),
         np.ascontiguousarray(mesh.face_verts, dtype=np.int32), np.ascontiguousarray(mesh.owner, dtype=np.int32),
         np.ascontiguousarray(mesh.neighbour, dtype=np.int32)]
    v = [C.c_int32(0) for _ in range(4)]
    st = lib.cpf_mesh_flags_host(_ptr(a[0]), mesh.n_points, _ptr(a[1]), _ptr(a[2]), mesh.n_faces, _ptr(a[3]), _ptr(a[4]),
                                 mesh.n_internal, mesh.n_cells, *[C.byref(k) for k in v])
    if st != L.CPF_OK:
        raise L.CpfError(st, "cpf_mesh_flags_host")
    return dict(all_hex=v[0].value, z_layered=v[1].value, z_thin=v[2].value, mixed=v[3].value)


def mesh_box_records_host(mesh):
    """The mesh's box records ([n_cells, 16] float64; ``cpf_mesh_box_records_host``), or None if some cell is not an axis-aligned
    box.  Layout: csrc/cpf_walk.h "box records"."""
    lib = L.load()
    a = [np.ascontiguousarray(mesh.points, dtype=np.float64), np.ascontiguousarray(mesh.face_offsets, dtype=np.int32),
         np.ascontiguousarray(mesh.face_verts, dtype=np.int32), np.ascontiguousarray(mesh.owner, dtype=np.int32),
         np.ascontiguousarray(mesh.neighbour, dtype=np.int32)]
    is_box = C.c_int32(0)
    rec = np.zeros((mesh.n_cells, 16), np.float64)
    st = lib.cpf_mesh_box_records_host(_ptr(a[0]), mesh.n_points, _ptr(a[1]), _ptr(a[2]), mesh.n_faces, _ptr(a[3]), _ptr(a[4]),
                                       mesh.n_internal, mesh.n_cells, C.byref(is_box), _ptr(rec))
    if st != L.CPF_OK:
        raise L.CpfError(st, "cpf_mesh_box_records_host")
    return rec if is_box.value else None


def pack_mesh_parts(parts):
    """(ctypes array of cpf_mesh_part, the numpy arrays it points into)."""
    arr = (L.MeshPart * len(parts))()
    keep = []
    for k, m in enumerate(parts):
        lab = np.int64 if np.asarray(m.owner).dtype == np.int64 else np.int32
        pts = np.ascontiguousarray(m.points, dtype=np.float64)
        fo = np.ascontiguousarray(m.face_offsets, dtype=lab); fv = np.ascontiguousarray(m.face_verts, dtype=lab)
        ow = np.ascontiguousarray(m.owner, dtype=lab); ne = np.ascontiguousarray(m.neighbour, dtype=lab)
        keep += [pts, fo, fv, ow, ne]
        arr[k] = L.MeshPart(pts.ctypes.data, pts.shape[0], fo.ctypes.data, fv.ctypes.data, ow.shape[0], ow.ctypes.data,
                            ne.ctypes.data if ne.size else None, ne.shape[0], int(m.n_cells), 8 if lab == np.int64 else 4)
    return arr, keep


def merge_mesh_parts(parts):
    """cpf_merge_mesh_parts on the host (no GPU): the stitched global mesh as a cases.PolyMesh (64-bit labels)."""
    from .cases.polymesh import PolyMesh
    lib = L.load()
    arr, keep = pack_mesh_parts(parts)
    h = C.c_void_p()
    r = lib.cpf_merge_mesh_parts(arr, len(parts), C.byref(h))
    if r != L.CPF_OK:
        raise L.CpfError(r, (lib.cpf_merge_last_error() or b"").decode())
    try:
        n = [C.c_int64() for _ in range(5)]
        lib.cpf_merged_mesh_sizes(h, *[C.byref(v) for v in n])
        n_points, n_faces, n_fv, n_int, n_cells = (v.value for v in n)
        pts = np.empty((n_points, 3)); fo = np.empty(n_faces + 1, np.int64); fv = np.empty(n_fv, np.int64)
        ow = np.empty(n_faces, np.int64); ne = np.empty(n_int, np.int64)
        lib.cpf_merged_mesh_copy(h, _ptr(pts), _ptr(fo), _ptr(fv), _ptr(ow), _ptr(ne))
    finally:
        lib.cpf_merged_mesh_free(h)
    return PolyMesh(points=pts, face_offsets=fo, face_verts=fv, owner=ow, neighbour=ne, n_cells=int(n_cells))


# ------------------------------------------------------------------------------------------------
# the fragments' call surface
# ------------------------------------------------------------------------------------------------
DICT_DEFAULTS = dict(                       # src/initCuda.H:49-57 (getOrDefault values)
    seedingBox=((0.0, 0.0, 0.0), (30.0, 30.0, 30.0)), numParticles=1000, startTime=0.0, endTime=1e05,
    dt=1e-4, diffusionCoeff=5.7e-6, saveInterval=10)


class CudaParticles:
    """What ``initCuda.H`` sets up and ``advect.H`` advances, for one solver process.

    ``writer(step, xyzw, vel, cell)`` is called with the cadence of
    ``writeParticles2VTU`` (frame 0 at init, then ``step % saveInterval == 0`` or at
    ``endTime``, src/initCuda.H:201, src/advect.H:166-169).
    """

    def __init__(self, mesh, U, particle_dict: Optional[dict] = None, device: int = 0,
                 writer: Optional[Callable] = None, positions: Optional[np.ndarray] = None, seed_order: int = 1):
        d = dict(DICT_DEFAULTS)
        d.update(particle_dict or {})
        self.numParticles = int(d["numParticles"])
        self.particleStartTime = float(d["startTime"])
        self.particleEndTime = float(d["endTime"])
        self.dt = float(d["dt"])
        self.diffusionCoeff = float(d["diffusionCoeff"])
        self.saveInterval = int(d["saveInterval"])
        self.seedingBox = d["seedingBox"]
        # hard-coded switches of the fragment (src/initCuda.H:64-72)
        self.usingAdvection = True
        self.usingBrownianMotion = True
        self.reflectWall = True
        self.step = 0
        self.writer = writer
        self.ctx = Context(device)
        if self.usingBrownianMotion and self.diffusionCoeff > 0:
            # diffusion scrambles the cell order ~5x faster than advection does (tools/sort_decay.py): shorter cadence
            self.ctx.set_option("sort_interval", 25)
        self.ctx.set_mesh(mesh)
        self.ctx.set_velocity(U)
        if positions is not None:            # parity runs inject positions
            self.ctx.set_particles(positions)
            self.numParticles = int(np.asarray(positions).shape[0])
        else:
            lo, hi = self.seedingBox
            self.ctx.seed_box(self.numParticles, lo, hi, seed_order)
        self.outOfDomain = self.ctx.locate_initial()       # RTQuery + cudaReportParticles
        self.ctx.sort_by_cell()
        if self.writer is not None:
            # frame 0 carries the velocities of one advect and out-of-domain particles as inactive, like the reference
            # (src/initCuda.H:184-201): a cycle of zero length (moves nothing, does not count as a step)
            self.ctx.step(0.0, 0.0, 1, L.STEP_STORE_VEL)
            self._write(0)

    def _flags(self, store_vel: bool) -> int:
        f = 0
        if not self.reflectWall:
            f |= L.STEP_NO_REFLECT
        if store_vel:
            f |= L.STEP_STORE_VEL
        return f

    def _write(self, frame: int):
        xyzw, cell, vel = self.ctx.get_particles(want_vel=True)
        self.writer(frame, xyzw, vel, cell)

    def advect(self, run_time_value: float, delta_t: float, U=None) -> int:
        """One ``#include "advect.H"``: returns the number of Lagrangian cycles done."""
        if not (self.particleStartTime <= run_time_value <= self.particleEndTime):     # advect.H:33
            return 0
        n_cycles = max(int(math.ceil(delta_t / self.dt)), 1)                           # advect.H:36
        cycle_dt = delta_t / n_cycles                                                  # advect.H:37
        if U is not None:                                                              # advect.H:44-57
            self.ctx.set_velocity(U)
        D = self.diffusionCoeff if self.usingBrownianMotion else 0.0
        done = 0
        while done < n_cycles:                                                         # advect.H:86
            will_write = self.writer is not None and (
                self.step % self.saveInterval == 0 or run_time_value == self.particleEndTime)
            if will_write:
                chunk = 1
            else:
                # cycles until the next output point run inside ONE launch (U is frozen during the loop)
                to_next = self.saveInterval - (self.step % self.saveInterval) if self.writer is not None else n_cycles
                chunk = max(1, min(n_cycles - done, to_next))
            flags = self._flags(will_write) | (L.STEP_FUSE_CYCLES if chunk > 1 else 0)
            self.ctx.step(cycle_dt, D, chunk, flags)
            if will_write:                                                             # advect.H:166-169
                self._write(self.step + 1)
            self.step += chunk                                                         # advect.H:182
            done += chunk
        return n_cycles

    def particles(self):
        return self.ctx.get_particles()

    def close(self):
        self.ctx.close()


class StagedCloud:
    """The reference's stage-by-stage call surface on its own array layouts (Particle = double4 AoS,
    vec4d disps / vels, int ids) -- cuda/common.h:32-77, query/ConvexQuery.h:33-46 -- over the
    ``cpf_stage_*`` entry points.  Method names are the reference wrappers' names; ids are CELL ids
    (reference: tet ids, cell = tet / 12).  Arrays live in device memory (``cpf_dev_alloc``); the
    numpy accessors copy."""

    def __init__(self, ctx: Context, n: int):
        self.ctx, self.n = ctx, int(n)
        self._bufs = {}
        for name, nbytes in (("P", 32 * n), ("vels", 32 * n), ("disps", 32 * n), ("ids", 4 * n)):
            p = C.c_void_p()
            ctx._ck(ctx.lib.cpf_dev_alloc(ctx.h, max(nbytes, 16), C.byref(p)))
            ctx._ck(ctx.lib.cpf_dev_memset(ctx.h, p, 0, max(nbytes, 16)))
            self._bufs[name] = p

    def close(self):
        for p in self._bufs.values():
            self.ctx.lib.cpf_dev_free(self.ctx.h, p)
        self._bufs = {}

    def _put(self, name, a):
        a = np.ascontiguousarray(a)
        self.ctx._ck(self.ctx.lib.cpf_copy_to_device(self.ctx.h, self._bufs[name], _ptr(a), a.nbytes))

    def _get(self, name, shape, dtype):
        a = np.empty(shape, dtype)
        self.ctx._ck(self.ctx.lib.cpf_copy_to_host(self.ctx.h, _ptr(a), self._bufs[name], a.nbytes))
        return a

    def set(self, xyzw=None, ids=None):
        if xyzw is not None:
            self._put("P", np.asarray(xyzw, np.float64).reshape(self.n, 4))
        if ids is not None:
            self._put("ids", np.asarray(ids, np.int32).reshape(self.n))

    particles = property(lambda s: s._get("P", (s.n, 4), np.float64))
    vels = property(lambda s: s._get("vels", (s.n, 4), np.float64))
    disps = property(lambda s: s._get("disps", (s.n, 4), np.float64))
    ids = property(lambda s: s._get("ids", (s.n,), np.int32))

    def cudaInitParticles(self, lower, upper, order: int = 1):            # cuda/particles.cu:100-108
        lo = np.ascontiguousarray(lower, dtype=np.float64); hi = np.ascontiguousarray(upper, dtype=np.float64)
        self.ctx._ck(self.ctx.lib.cpf_stage_seed_box(self.ctx.h, self._bufs["P"], self.n, _ptr(lo), _ptr(hi), order))

    def RTQuery(self):                                                    # query/RTQuery.cu:295-310
        self.ctx._ck(self.ctx.lib.cpf_stage_locate_initial(self.ctx.h, self._bufs["P"], self._bufs["ids"], self.n))

    def cudaReportParticles(self) -> int:                                 # cuda/particles.cu:763-775
        out = C.c_int64()
        self.ctx._ck(self.ctx.lib.cpf_stage_count_outside(self.ctx.h, self._bufs["ids"], self.n, C.byref(out)))
        return out.value

    def cudaAdvect(self, dt: float, mode: str = "TetVelocity"):           # cuda/particles.cu:403-448
        b = self._bufs
        if mode == "TetVelocity":
            self.ctx._ck(self.ctx.lib.cpf_stage_advect(self.ctx.h, b["P"], b["ids"], b["vels"], b["disps"], dt, self.n))
        elif mode == "VertexVelocity":                                    # :428-437 -> particleAdvectKernel :244-313
            self.ctx._ck(self.ctx.lib.cpf_stage_advect_vertex(self.ctx.h, b["P"], b["ids"], b["vels"], b["disps"], dt, self.n))
        elif mode == "ConstantVelocity":                                  # :439-445 -> particleAdvectConstVel :376-399
            self.ctx._ck(self.ctx.lib.cpf_stage_advect_const(self.ctx.h, b["P"], b["ids"], b["vels"], b["disps"], dt, self.n))
        else:
            raise ValueError("cudaAdvect: mode must be TetVelocity, VertexVelocity or ConstantVelocity (cuda/particles.cu:417-445)")

    def cudaBrownianMotion(self, dt: float, D: float, step: int):         # cuda/particles.cu:577-599
        b = self._bufs
        self.ctx._ck(self.ctx.lib.cpf_stage_brownian(self.ctx.h, b["P"], b["disps"], dt, self.n, D, step))

    def convexTetQuery(self):                                             # query/ConvexQuery.cu:218-234
        b = self._bufs
        self.ctx._ck(self.ctx.lib.cpf_stage_locate(self.ctx.h, b["P"], b["disps"], b["ids"], self.n))

    def convexWallReflect(self):                                          # query/ConvexQuery.cu:438-458
        b = self._bufs
        self.ctx._ck(self.ctx.lib.cpf_stage_reflect(self.ctx.h, b["ids"], b["P"], b["vels"], b["disps"], self.n))

    def cudaMoveParticles(self):                                          # cuda/particles.cu:706-716
        b = self._bufs
        self.ctx._ck(self.ctx.lib.cpf_stage_move(self.ctx.h, b["P"], b["disps"], self.n))
