"""ctypes binding of ``libcudaParticleAdvection.so`` -- exactly the symbols ``include/cpf.h`` declares.

There is NO fallback: if the library is missing, or the box has no HIP device, the product
path raises.  (The CPU restatements under ``oracle/`` are test infrastructure and are never
imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import re

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "lib", "libcudaParticleAdvection.so")
HEADER_PATH = os.path.join(os.path.dirname(PKG_DIR), "include", "cpf.h")

CPF_OK, CPF_ERR_ARG, CPF_ERR_STATE, CPF_ERR_MESH, CPF_ERR_HIP, CPF_ERR_NOMEM, CPF_WARN_NAN = range(7)
CELL_LOST, CELL_FROZEN = -1, -2
STEP_DEFAULT, STEP_NO_REFLECT, STEP_STORE_VEL, STEP_FUSE_CYCLES, STEP_VERTEX_VELOCITY = 0, 1, 2, 4, 8
HANDOFF_DOUBLES = 5
MAX_RANKS, COMM_ID_BYTES, COMM_RCCL, COMM_INPROCESS = 64, 128, 1, 2


class CpfError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__("cpf status %d: %s" % (status, message))
        self.status = status


class LibraryMissing(ImportError):
    pass


_vp, _i64, _i32, _u32, _dbl, _int = C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_double, C.c_int
_ctx = C.c_void_p

class MeshPart(C.Structure):
    """cpf_mesh_part (include/cpf.h): one rank's piece of a decomposed polyMesh."""
    _fields_ = [("points", _vp), ("nPoints", _i64), ("faceOffsets", _vp), ("faceVerts", _vp), ("nFaces", _i64),
                ("owner", _vp), ("neighbour", _vp), ("nInternalFaces", _i64), ("nCells", _i64), ("labelBytes", _int)]


# cpf_comm (include/cpf.h): the three collectives of a hand-off, on device memory, asynchronous on a stream
ALL_GATHER_FN = C.CFUNCTYPE(_int, _vp, _vp, _vp, C.c_size_t, _vp)
ALL_REDUCE_FN = C.CFUNCTYPE(_int, _vp, _vp, C.c_size_t, _vp)
ALL_TO_ALL_V_FN = C.CFUNCTYPE(_int, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64), _vp, C.POINTER(_i64), C.POINTER(_i64), _vp)
COMM_DESTROY_FN = C.CFUNCTYPE(None, _vp)
COMM_ERROR_FN = C.CFUNCTYPE(C.c_char_p, _vp)


class Comm(C.Structure):
    _fields_ = [("self", _vp), ("rank", _int), ("nRanks", _int), ("all_gather", ALL_GATHER_FN),
                ("all_reduce_sum_f64", ALL_REDUCE_FN), ("all_to_all_v", ALL_TO_ALL_V_FN), ("destroy", COMM_DESTROY_FN),
                ("last_error", COMM_ERROR_FN)]


class ShardStats(C.Structure):
    """cpf_shard_stats (include/cpf.h)"""
    _fields_ = [("n", _i64), ("capacity", _i64), ("stepIndex", _i64), ("particleSteps", _i64), ("handedOff", _i64),
                ("exchanges", _i64), ("rebalances", _i64), ("grown", _i64), ("sendGrown", _i64), ("kernelLaunches", _i64),
                ("kernelMs", _dbl), ("handoffHostMs", _dbl), ("handoffWaitMs", _dbl), ("hostWorkMsPerHandoff", _dbl),
                ("commDeviceMs", _dbl), ("commEvents", _i64), ("overlapDepth", _i32), ("nRanks", _i32), ("rank", _i32)]


_shard = C.c_void_p
# the cpf_shard_* entry points: exported by the product library AND, over a host-memory stand-in device, by the tests' own
# tests/host_shard/libcpf_shard_host.so (see bind_shard_signatures)
SHARD_SIGNATURES = {
    "cpf_shard_create": (_int, [_ctx, C.POINTER(Comm), _i64, _vp, C.POINTER(_shard)]),
    "cpf_shard_destroy": (_int, [_shard]),
    "cpf_shard_last_error": (C.c_char_p, [_shard]),
    "cpf_shard_set_option": (_int, [_shard, C.c_char_p, _dbl]),
    "cpf_shard_set_particles_dev": (_int, [_shard, _vp, _vp, _vp, _vp, _vp, _i64, _i64]),
    "cpf_shard_seed_box": (_int, [_shard, _i64, _vp, _vp, _int, C.POINTER(_i64)]),
    "cpf_shard_step": (_int, [_shard, _dbl, _dbl, _int, C.c_uint]),
    "cpf_shard_flush": (_int, [_shard]),
    "cpf_shard_exchange": (_int, [_shard]),
    "cpf_shard_rebalance": (_int, [_shard]),
    "cpf_shard_sort": (_int, [_shard]),
    "cpf_shard_set_velocity": (_int, [_shard, _vp, _i64]),
    "cpf_shard_set_velocity_slice": (_int, [_shard, _vp, _i64]),
    "cpf_shard_arrays": (_int, [_shard, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                C.POINTER(_i64), C.POINTER(_i64)]),
    "cpf_shard_get_local": (_int, [_shard, _vp, _vp, _vp, _vp, _vp]),
    "cpf_shard_global_count": (_int, [_shard, C.POINTER(_i64)]),
    "cpf_shard_cell_ranges": (_int, [_shard, _vp]),
    "cpf_shard_gather": (_int, [_shard, _int, _vp, _vp, _vp]),
    "cpf_shard_write_vtu": (_int, [_shard, _int, C.c_char_p, C.POINTER(_dbl)]),
    "cpf_shard_write_vtu_wait": (_int, [_shard]),
    "cpf_shard_get_stats": (_int, [_shard, C.POINTER(ShardStats)]),
}


def bind_shard_signatures(lib):
    for name, (res, args) in SHARD_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


# name -> (restype, argtypes); mirrors include/cpf.h one to one
SIGNATURES = {
    "cpf_abi_version": (_int, []),
    "cpf_create": (_int, [_int, C.POINTER(_ctx)]),
    "cpf_destroy": (_int, [_ctx]),
    "cpf_last_error": (C.c_char_p, [_ctx]),
    "cpf_set_stream": (_int, [_ctx, _vp]),
    "cpf_use_own_stream": (_int, [_ctx]),
    "cpf_synchronize": (_int, [_ctx]),
    "cpf_set_mesh": (_int, [_ctx, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64]),
    "cpf_set_mesh_l64": (_int, [_ctx, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64]),
    "cpf_merge_mesh_parts": (_int, [C.POINTER(MeshPart), _int, C.POINTER(_vp)]),
    "cpf_merge_last_error": (C.c_char_p, []),
    "cpf_merged_mesh_sizes": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64),
                                     C.POINTER(_i64)]),
    "cpf_merged_mesh_copy": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "cpf_merged_mesh_free": (None, [_vp]),
    "cpf_set_mesh_parts": (_int, [_ctx, C.POINTER(MeshPart), _int]),
    "cpf_mesh_info": (_int, [_ctx, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "cpf_get_mesh_tables": (_int, [_ctx, _vp, _vp, _vp]),
    "cpf_get_mesh_groups": (_int, [_ctx, C.POINTER(_i64), C.POINTER(_i64), _vp, _vp]),
    "cpf_get_mesh_flags": (_int, [_ctx, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cpf_build_mesh_tables_host": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64, C.POINTER(_i64), C.POINTER(_i64),
                                          C.POINTER(_i64), _vp, _vp, _vp, _vp, _vp]),
    "cpf_mesh_flags_host": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cpf_mesh_box_records_host": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64, C.POINTER(C.c_int32), _vp]),
    "cpf_set_velocity": (_int, [_ctx, _vp, _i64]),
    "cpf_set_velocity_dev": (_int, [_ctx, _vp, _i64]),
    "cpf_alloc_particles": (_int, [_ctx, _i64]),
    "cpf_seed_box": (_int, [_ctx, _i64, _vp, _vp, _int]),
    "cpf_set_particles": (_int, [_ctx, _i64, _vp, _vp]),
    "cpf_locate_initial": (_int, [_ctx, C.POINTER(_i64)]),
    "cpf_step": (_int, [_ctx, _dbl, _dbl, _int, C.c_uint]),
    "cpf_sort_by_cell": (_int, [_ctx]),
    "cpf_num_particles": (_int, [_ctx, C.POINTER(_i64)]),
    "cpf_get_particles": (_int, [_ctx, _vp, _vp, _vp]),
    "cpf_get_counters": (_int, [_ctx, _vp]),
    "cpf_set_seed": (_int, [_ctx, _u32]),
    "cpf_set_option": (_int, [_ctx, C.c_char_p, _dbl]),
    "cpf_step_kernel_name": (_int, [_ctx, _dbl, C.c_uint, C.c_char_p, C.c_size_t]),
    "cpf_step_dev": (_int, [_ctx, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _u32, _int, C.c_uint]),
    "cpf_locate_initial_dev": (_int, [_ctx, _vp, _vp, _vp, _vp, _i64]),
    "cpf_seed_box_dev": (_int, [_ctx, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _int]),
    "cpf_sort_by_cell_dev": (_int, [_ctx, _vp, _vp, _vp, _vp, _vp, _i64]),
    "cpf_sort_by_cell_dev_to": (_int, [_ctx, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "cpf_pack_leavers_dev": (_int, [_ctx, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _int, _int, _vp, _i64, _vp, _vp]),
    "cpf_cell_histogram_dev": (_int, [_ctx, _vp, _i64, _dbl, _vp]),
    "cpf_cell_ranges_dev": (_int, [_ctx, _vp, _int, _vp]),
    "cpf_unpack_arrivals_dev": (_int, [_ctx, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64]),
    "cpf_dev_alloc": (_int, [_ctx, C.c_size_t, C.POINTER(_vp)]),
    "cpf_dev_free": (_int, [_ctx, _vp]),
    "cpf_dev_memset": (_int, [_ctx, _vp, _int, C.c_size_t]),
    "cpf_copy_to_device": (_int, [_ctx, _vp, _vp, C.c_size_t]),
    "cpf_copy_to_host": (_int, [_ctx, _vp, _vp, C.c_size_t]),
    "cpf_copy_dev": (_int, [_ctx, _vp, _vp, C.c_size_t]),
    "cpf_stage_seed_box": (_int, [_ctx, _vp, _i64, _vp, _vp, _int]),
    "cpf_stage_locate_initial": (_int, [_ctx, _vp, _vp, _i64]),
    "cpf_stage_count_outside": (_int, [_ctx, _vp, _i64, C.POINTER(_i64)]),
    "cpf_stage_advect": (_int, [_ctx, _vp, _vp, _vp, _vp, _dbl, _i64]),
    "cpf_set_tets": (_int, [_ctx, _vp, _i64, _vp, _i64, _int]),
    "cpf_set_vertex_velocity": (_int, [_ctx, _vp, _i64]),
    "cpf_stage_advect_vertex": (_int, [_ctx, _vp, _vp, _vp, _vp, _dbl, _i64]),
    "cpf_stage_advect_const": (_int, [_ctx, _vp, _vp, _vp, _vp, _dbl, _i64]),
    "cpf_stage_brownian": (_int, [_ctx, _vp, _vp, _dbl, _i64, _dbl, _u32]),
    "cpf_stage_locate": (_int, [_ctx, _vp, _vp, _vp, _i64]),
    "cpf_stage_reflect": (_int, [_ctx, _vp, _vp, _vp, _vp, _i64]),
    "cpf_stage_move": (_int, [_ctx, _vp, _vp, _i64]),
    "cpf_write_vtu": (_int, [_ctx, C.c_char_p, C.POINTER(_dbl)]),
    "cpf_write_vtu_async": (_int, [_ctx, C.c_char_p, C.POINTER(_dbl)]),
    "cpf_write_vtu_wait": (_int, [_ctx]),
    "cpf_write_vtu_arrays": (_int, [C.c_char_p, _i64, _vp, _vp, _vp, C.POINTER(_dbl)]),
    "cpf_write_vtu_arrays_binary": (_int, [C.c_char_p, _i64, _vp, _vp, _vp, C.POINTER(_dbl)]),
    "cpf_timing_enable": (_int, [_ctx, _int]),
    "cpf_timing_read": (_int, [_ctx, C.POINTER(_i64), C.POINTER(_dbl)]),
    "cpf_timing_poll": (_int, [_ctx, C.POINTER(_i64), C.POINTER(_dbl)]),
    "cpf_traj_create": (_int, [C.POINTER(_vp)]),
    "cpf_traj_destroy": (None, [_vp]),
    "cpf_traj_add": (_int, [_ctx, _vp]),
    "cpf_traj_add_stage": (_int, [_ctx, _vp, _vp, _i64]),
    "cpf_traj_add_host": (_int, [_vp, _vp, _i64]),
    "cpf_traj_sizes": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "cpf_traj_save_obj": (_int, [_vp, C.c_char_p]),
    "cpf_traj_write_vtk": (_int, [_vp, C.c_char_p]),
    "cpf_traj_save_obj_arrays": (_int, [C.c_char_p, _i64, _vp, _vp]),
    "cpf_traj_write_vtk_arrays": (_int, [C.c_char_p, _i64, _vp, _vp]),
    "cpf_device_count": (_int, [C.POINTER(_int)]),
    "cpf_comm_unique_id": (_int, [_vp, _int]),
    "cpf_comm_default_kind": (_int, []),
    "cpf_comm_create": (_int, [_vp, _int, _int, _int, C.POINTER(C.POINTER(Comm))]),
    "cpf_comm_destroy": (None, [C.POINTER(Comm)]),
    "cpf_comm_last_error": (C.c_char_p, []),
}
SIGNATURES.update(SHARD_SIGNATURES)

_lib = None


def header_symbols(path: str = HEADER_PATH):
    """Function names declared in include/cpf.h (used by the symbol-export test)."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cpf_[a-z0-9_]+)\s*\(", text)))


def load() -> C.CDLL:
    """Load the HIP library or fail loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or make -C cudaparticlesfoam_amd/csrc). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here == header/library mismatch: loud by design
        fn.restype = res
        fn.argtypes = args
    if lib.cpf_abi_version() != 1:
        raise LibraryMissing("libcudaParticleAdvection.so ABI version mismatch")
    _lib = lib
    return lib


def check(lib, ctx, status: int) -> None:
    if status != CPF_OK:
        msg = lib.cpf_last_error(ctx)
        raise CpfError(status, (msg or b"").decode("utf-8", "replace"))
