"""Test infrastructure for the CPU (gloo) tests of the sharded cloud.

The product's hand-off logic (cudaparticlesfoam_amd/csrc/cpf_shard_core.h) is compiled a second time, by
tests/host_shard/Makefile, over a host-memory stand-in for the device whose step is the CPU checker
(tests/host_shard/host_shard.cpp): `HostCase` is that library's "context", `GlooComm` a `cpf_comm` whose three
collectives are torch.distributed calls on host memory.  Together they let `parallel.ShardedCloud` -- the same binding
the GPU path uses -- run with world size 2 / 3 / 8 on a machine without a GPU.  The product has no CPU path.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

from cudaparticlesfoam_amd import _lib as L

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "host_shard", "libcpf_shard_host.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-C", os.path.join(HERE, "host_shard"), "-s"], check=True)
        lib = C.CDLL(LIB_PATH)
        L.bind_shard_signatures(lib)
        lib.cpf_host_case_create.restype = C.c_void_p
        lib.cpf_host_case_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_int64, C.c_void_p, C.c_int64, C.c_double]
        lib.cpf_host_case_destroy.argtypes = [C.c_void_p]
        lib.cpf_host_case_timing.argtypes = [C.c_void_p, C.c_int]
        lib.cpf_host_case_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        _lib = lib
    return _lib


class HostCase:
    """What `ShardedCloud` takes for a context on the host stand-in: `.h` (the handle) and the few context calls
    bench.run() makes besides the shard's own."""

    def __init__(self, tables, U, fake_ms_per_particle=2.0e-8):
        self.lib = load()
        t = tables
        U = np.ascontiguousarray(U, dtype=np.float64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)     # noqa: E731
        off = np.ascontiguousarray(t.cell_off, np.int32); planes = np.ascontiguousarray(t.planes, np.float64)
        nbr = np.ascontiguousarray(t.nbr, np.int32); goff = np.ascontiguousarray(t.group_off, np.int32)
        gnbr = np.ascontiguousarray(t.group_nbr, np.int32)
        self.n_cells = int(t.n_cells)
        self.h = C.c_void_p(self.lib.cpf_host_case_create(p(off), p(planes), p(nbr), nbr.shape[0], p(goff), goff.shape[0] - 1,
                                                          p(gnbr), int(goff[-1]), p(U), self.n_cells, float(fake_ms_per_particle)))

    def close(self):
        for sh in list(getattr(self, "_shards", [])):
            sh.close()
        if self.h:
            self.lib.cpf_host_case_destroy(self.h)
            self.h = None

    # -- the context calls of bench.run()
    def set_option(self, key, value):
        pass

    def counters(self):
        return {"particle_steps": 0, "cells_visited": 0, "reflections": 0, "lost": 0}

    def timing_enable(self, on=True):
        self.lib.cpf_host_case_timing(self.h, 1 if on else 0)

    def timing_read(self):
        a, b = C.c_int64(), C.c_double()
        self.lib.cpf_host_case_timing_read(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def step_launches(self):
        self.lib.cpf_host_case_step_launches.restype = C.c_int64
        self.lib.cpf_host_case_step_launches.argtypes = [C.c_void_p]
        return int(self.lib.cpf_host_case_step_launches(self.h))

    def step_kernel_name(self, D=0.0, flags=0):
        return "tests/host_shard (CPU checker stand-in)"

    def mesh_info(self):
        return {"n_cells": self.n_cells}


def _view(ptr, nbytes, dtype=np.uint8):
    if not nbytes:
        return np.empty(0, dtype)
    return np.frombuffer((C.c_char * int(nbytes)).from_address(int(ptr)), dtype=dtype)


class GlooComm:
    """A `cpf_comm` (include/cpf.h) over torch.distributed on HOST memory: what a host with its own transport would fill in."""

    def __init__(self, dist, group=None):
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.calls = dict(all_gather=0, all_reduce=0, all_to_all_v=0)
        self._err = b""

        def all_gather(_self, send, recv, nbytes, _stream):
            try:
                self.calls["all_gather"] += 1
                s = torch.from_numpy(_view(send, nbytes).copy())
                out = torch.from_numpy(_view(recv, nbytes * self.world))
                dist.all_gather(list(out.view(self.world, -1).unbind(0)), s, group=group)
                return L.CPF_OK
            except Exception as e:                      # noqa: BLE001
                self._err = repr(e).encode()
                return L.CPF_ERR_STATE

        def all_reduce(_self, buf, count, _stream):
            try:
                self.calls["all_reduce"] += 1
                t = torch.from_numpy(_view(buf, count * 8, np.float64))
                dist.all_reduce(t, group=group)
                return L.CPF_OK
            except Exception as e:                      # noqa: BLE001
                self._err = repr(e).encode()
                return L.CPF_ERR_STATE

        def all_to_all_v(_self, send, s_off, s_bytes, recv, r_off, r_bytes, _stream):
            try:
                self.calls["all_to_all_v"] += 1
                W = self.world
                so, sb = [s_off[r] for r in range(W)], [s_bytes[r] for r in range(W)]
                ro, rb = [r_off[r] for r in range(W)], [r_bytes[r] for r in range(W)]
                parts = [_view(send + so[r], sb[r]).copy() if sb[r] else np.empty(0, np.uint8) for r in range(W)]
                inp = torch.from_numpy(np.concatenate(parts)) if sum(sb) else torch.empty(0, dtype=torch.uint8)
                out = torch.empty(sum(rb), dtype=torch.uint8)
                dist.all_to_all_single(out, inp, rb, sb, group=group)
                o = out.numpy(); at = 0
                for r in range(W):
                    if rb[r]:
                        _view(recv + ro[r], rb[r])[:] = o[at:at + rb[r]]
                    at += rb[r]
                return L.CPF_OK
            except Exception as e:                      # noqa: BLE001
                self._err = repr(e).encode()
                return L.CPF_ERR_STATE

        self._keep = (L.ALL_GATHER_FN(all_gather), L.ALL_REDUCE_FN(all_reduce), L.ALL_TO_ALL_V_FN(all_to_all_v),
                      L.COMM_ERROR_FN(lambda _self: self._err))
        self.struct = L.Comm(None, self.rank, self.world, self._keep[0], self._keep[1], self._keep[2], L.COMM_DESTROY_FN(),
                             self._keep[3])
        self.ptr = C.pointer(self.struct)


class ThreadGroup:
    """What the ranks of a `ThreadComm` share: a barrier and one slot per rank."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def abort(self):
        self.barrier.abort()


class ThreadComm:
    """A `cpf_comm` whose ranks are THREADS of this process over host memory (a barrier and per-rank slots): the randomised
    test of the shard logic plays many worlds in one process this way, without a process group per case."""

    def __init__(self, group, rank):
        self.group, self.rank, self.world = group, rank, group.world
        self._err = b""
        g = group

        def fail(e):
            self._err = repr(e).encode()
            g.abort()
            return L.CPF_ERR_STATE

        def all_gather(_self, send, recv, nbytes, _stream):
            try:
                g.slots[rank] = _view(send, nbytes).copy()
                g.barrier.wait()
                _view(recv, nbytes * g.world)[:] = np.concatenate([g.slots[r] for r in range(g.world)]) if nbytes else 0
                g.barrier.wait()
                return L.CPF_OK
            except Exception as e:                      # noqa: BLE001
                return fail(e)

        def all_reduce(_self, buf, count, _stream):
            try:
                g.slots[rank] = _view(buf, count * 8, np.float64).copy()
                g.barrier.wait()
                acc = g.slots[0].copy()
                for r in range(1, g.world):
                    acc += g.slots[r]                   # (rank order: the same sum on every rank)
                g.barrier.wait()
                _view(buf, count * 8, np.float64)[:] = acc
                return L.CPF_OK
            except Exception as e:                      # noqa: BLE001
                return fail(e)

        def all_to_all_v(_self, send, s_off, s_bytes, recv, r_off, r_bytes, _stream):
            try:
                W = g.world
                g.slots[rank] = [_view(send + s_off[r], s_bytes[r]).copy() if s_bytes[r] else None for r in range(W)]
                g.barrier.wait()
                for r in range(W):
                    if r_bytes[r]:
                        part = g.slots[r][rank]
                        assert part is not None and part.size == r_bytes[r], "send and receive sizes disagree"
                        _view(recv + r_off[r], r_bytes[r])[:] = part
                g.barrier.wait()
                return L.CPF_OK
            except Exception as e:                      # noqa: BLE001
                return fail(e)

        self._keep = (L.ALL_GATHER_FN(all_gather), L.ALL_REDUCE_FN(all_reduce), L.ALL_TO_ALL_V_FN(all_to_all_v),
                      L.COMM_ERROR_FN(lambda _self: self._err))
        self.struct = L.Comm(None, rank, g.world, self._keep[0], self._keep[1], self._keep[2], L.COMM_DESTROY_FN(), self._keep[3])
        self.ptr = C.pointer(self.struct)


def cloud(case, cell_lo, capacity, comm=None, **kw):
    from cudaparticlesfoam_amd.parallel import ShardedCloud
    return ShardedCloud(case, cell_lo, capacity, comm, lib=load(), **kw)
