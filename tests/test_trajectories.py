"""Trajectory collection and its two writers (addToTrajectories / saveTrajectories / writeStreamline2VTK,
cuda/utils.cpp:7-94): the library's files against the reference's own functions (oracle/_ref), byte for byte.  The writers are
host code: no GPU needed for the host-array entry points."""
import ctypes as C
import os

import numpy as np
import pytest


def _samples(n=37, steps=6, seed=3):
    """positions drifting over a few instants; some particles inactive from the start, some dying on the way, one alive once"""
    rng = np.random.default_rng(seed)
    P = np.zeros((n, 4)); P[:, :3] = rng.normal(size=(n, 3)) * [1.0, 1e-3, 1e4]; P[:, 3] = 1
    P[5, 3] = 0; P[11, 3] = 0
    out = []
    for k in range(steps):
        Q = P.copy()
        Q[:, :3] += k * rng.normal(size=(n, 3)) * 0.1
        if k >= 1:
            Q[20, 3] = 0                     # one sample only: left out of both files
        if k >= 3:
            Q[7, 3] = 0
        out.append(Q)
    out[2][3, :3] = [1e-7, -123456.789, 3.0]          # exponent and many-digit formatting
    return out


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref",
                                                    "libref_rtxadvect.so")), reason="oracle/_ref not built (needs the reference tree)")
def test_trajectory_files_equal_the_reference_writers(tmp_path, oracle_libs):
    from cudaparticlesfoam_amd import _lib as L
    lib = L.load()
    samples = _samples()
    ref = oracle_libs.RefLib()
    ref.trajectories(samples, str(tmp_path / "ref.obj"), str(tmp_path / "ref.vtk"))
    t = C.c_void_p()
    assert lib.cpf_traj_create(C.byref(t)) == L.CPF_OK
    for P in samples:
        P = np.ascontiguousarray(P)
        assert lib.cpf_traj_add_host(t, P.ctypes.data_as(C.c_void_p), P.shape[0]) == L.CPF_OK
    nt, npnt = C.c_int64(), C.c_int64()
    assert lib.cpf_traj_sizes(t, C.byref(nt), C.byref(npnt)) == L.CPF_OK
    alive = sum(int((P[:, 3] != 0).sum()) for P in samples)
    assert nt.value == samples[0].shape[0] and npnt.value == alive
    assert lib.cpf_traj_save_obj(t, str(tmp_path / "cpf.obj").encode()) == L.CPF_OK
    assert lib.cpf_traj_write_vtk(t, str(tmp_path / "cpf.vtk").encode()) == L.CPF_OK
    for ext in ("obj", "vtk"):
        a, b = open(tmp_path / ("ref." + ext), "rb").read(), open(tmp_path / ("cpf." + ext), "rb").read()
        assert len(a) > 500 and a == b, ext
    # the same writers on a host's own storage (what the name-compatible shims in compat/cuda/common.h call)
    off, xyz = [0], []
    for i in range(samples[0].shape[0]):
        pts = [P[i, :3] for P in samples if P[i, 3] != 0]
        xyz += pts; off.append(off[-1] + len(pts))
    off = np.asarray(off, np.int64); xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    assert lib.cpf_traj_save_obj_arrays(str(tmp_path / "arr.obj").encode(), len(off) - 1, p(off), p(xyz)) == L.CPF_OK
    assert lib.cpf_traj_write_vtk_arrays(str(tmp_path / "arr.vtk").encode(), len(off) - 1, p(off), p(xyz)) == L.CPF_OK
    for ext in ("obj", "vtk"):
        assert open(tmp_path / ("ref." + ext), "rb").read() == open(tmp_path / ("arr." + ext), "rb").read()
    # a second sample count is refused (the first sample fixes the particle count, cuda/utils.cpp:10-11)
    Q = np.zeros((5, 4))
    assert lib.cpf_traj_add_host(t, Q.ctypes.data_as(C.c_void_p), 5) == L.CPF_ERR_ARG
    lib.cpf_traj_destroy(t)


def test_empty_collection_writes_headers_only(tmp_path):
    from cudaparticlesfoam_amd import _lib as L
    lib = L.load()
    t = C.c_void_p()
    assert lib.cpf_traj_create(C.byref(t)) == L.CPF_OK
    assert lib.cpf_traj_save_obj(t, str(tmp_path / "e.obj").encode()) == L.CPF_OK
    assert lib.cpf_traj_write_vtk(t, str(tmp_path / "e.vtk").encode()) == L.CPF_OK
    assert open(tmp_path / "e.obj").read() == ""
    txt = open(tmp_path / "e.vtk").read()
    assert "POINTS 0 float" in txt and "LINES 0 0" in txt and "StreamlineID 1 0 int" in txt
    lib.cpf_traj_destroy(t)


@pytest.mark.gpu
def test_trajectories_of_the_context_cloud_and_of_the_staged_arrays(tmp_path, pitz, gpu_ctx_factory):
    """cpf_traj_add samples the context-owned cloud in particle-id order (whatever the re-sorts did to the arrays);
    cpf_traj_add_stage samples the reference's AoS particle array in device memory."""
    from cudaparticlesfoam_amd import _lib as L
    ctx = gpu_ctx_factory()
    ctx.set_mesh(pitz["mesh"]); ctx.set_velocity(pitz["U_analytic"])
    pz = pitz["pz"]
    ctx.seed_box(4000, (-0.0215, 0.03, 0.0001), (0.0, 0.0, -0.0001), 1)       # sticks out of the inlet: some are inactive
    n_out = ctx.locate_initial()
    ctx.set_option("sort_interval", 3)
    lib = ctx.lib
    t = C.c_void_p(); lib.cpf_traj_create(C.byref(t))
    snaps = []
    ctx.step(0.0, 0.0, 1)                                                     # frame-0 idiom: out-of-domain particles become inactive
    for k in range(4):
        assert lib.cpf_traj_add(ctx.h, t) == L.CPF_OK
        snaps.append(ctx.get_particles()[0].copy())
        ctx.step(1e-4, 0.0, 5)
    nt, npnt = C.c_int64(), C.c_int64()
    lib.cpf_traj_sizes(t, C.byref(nt), C.byref(npnt))
    assert nt.value == 4000 and npnt.value == sum(int((s[:, 3] != 0).sum()) for s in snaps) and 0 < n_out < 4000
    assert lib.cpf_traj_write_vtk(t, str(tmp_path / "g.vtk").encode()) == L.CPF_OK
    t2 = C.c_void_p(); lib.cpf_traj_create(C.byref(t2))
    for s in snaps:
        s = np.ascontiguousarray(s)
        lib.cpf_traj_add_host(t2, s.ctypes.data_as(C.c_void_p), s.shape[0])
    lib.cpf_traj_write_vtk(t2, str(tmp_path / "h.vtk").encode())
    assert open(tmp_path / "g.vtk", "rb").read() == open(tmp_path / "h.vtk", "rb").read()
    # staged path: the AoS array in device memory
    d = C.c_void_p()
    ctx._ck(lib.cpf_dev_alloc(ctx.h, snaps[0].nbytes, C.byref(d)))
    ctx._ck(lib.cpf_copy_to_device(ctx.h, d, snaps[1].ctypes.data_as(C.c_void_p), snaps[1].nbytes))
    t3 = C.c_void_p(); lib.cpf_traj_create(C.byref(t3))
    assert lib.cpf_traj_add_stage(ctx.h, t3, d, 4000) == L.CPF_OK
    lib.cpf_traj_sizes(t3, C.byref(nt), C.byref(npnt))
    assert npnt.value == int((snaps[1][:, 3] != 0).sum())
    ctx._ck(lib.cpf_dev_free(ctx.h, d))
    for h in (t, t2, t3):
        lib.cpf_traj_destroy(h)
