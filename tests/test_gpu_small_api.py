"""GPU: the small entry points of include/cpf.h that no other test calls by name."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_count_alloc_and_device_copy(gpu_ctx_factory, pitz):
    import torch
    from cudaparticlesfoam_amd import _lib as L
    lib = L.load()
    n_dev = C.c_int(-1)
    assert lib.cpf_device_count(C.byref(n_dev)) == L.CPF_OK and n_dev.value >= 1
    assert lib.cpf_device_count(None) == L.CPF_ERR_ARG
    ctx = gpu_ctx_factory()
    ctx.set_mesh(pitz["mesh"]); ctx.set_velocity(pitz["U_analytic"])
    # the reference's cudaMalloc block (src/initCuda.H:141-150): capacity checked, then a cloud that fits is seeded into it
    for bad in (0, -5, 1 << 31):
        assert lib.cpf_alloc_particles(ctx.h, bad) == L.CPF_ERR_ARG
    assert b"capacity" in lib.cpf_last_error(ctx.h)
    assert lib.cpf_alloc_particles(ctx.h, 5000) == L.CPF_OK
    ctx.seed_box(3000, *pitz["pz"].DOMAIN_BOX, 1)
    ctx.locate_initial()
    ctx.step(1e-4, 0.0, 3)
    xyzw, cell = ctx.get_particles()
    assert xyzw.shape == (3000, 4) and (cell >= 0).sum() > 2000
    # cpf_copy_dev: device to device on the context's stream
    a = torch.arange(1000, dtype=torch.float64, device="cuda")
    b = torch.zeros_like(a)
    torch.cuda.synchronize()
    assert lib.cpf_copy_dev(ctx.h, C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), 8000) == L.CPF_OK
    ctx.synchronize()
    assert torch.equal(a, b)
    assert lib.cpf_copy_dev(ctx.h, None, C.c_void_p(a.data_ptr()), 8) == L.CPF_ERR_ARG
