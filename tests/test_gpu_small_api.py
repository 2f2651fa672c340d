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


@pytest.mark.parametrize("D", [0.0, 2e-5])
def test_tuning_knobs_of_the_streaming_kernel_never_change_results(D, gpu_ctx_factory, pitz):
    """include/cpf.h: "tuning knobs, never semantics".  Tiles per chunk, the tail fraction, waves per CU and the time-stamp stride
    change how the persistent waves are dealt their work and how launches are timed -- positions and cells stay the same bits
    (with and without the kick, statistics and timing on)."""
    pz = pitz["pz"]
    n = 150_000
    xyz = pz.uniform_points(11, n, *pz.DOMAIN_BOX)

    def run(opts):
        ctx = gpu_ctx_factory()
        ctx.set_mesh(pitz["mesh"]); ctx.set_velocity(pitz["U_analytic"])
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_particles(xyz); ctx.locate_initial(); ctx.sort_by_cell()
        if "timing_stride" in opts:
            ctx.timing_enable(True)
        ctx.step(1e-4, D, 6)
        ctx.step(1e-4, D, 5, 4)                                   # CPF_STEP_FUSE_CYCLES
        if "timing_stride" in opts:
            launches, ms = ctx.timing_read()
            assert 0 < launches <= 7 and ms > 0
        return ctx.get_particles()

    want = run({})
    for opts in ({"stream_tiles_per_chunk": 1}, {"stream_tiles_per_chunk": 7}, {"stream_tiles_per_chunk": 1024},
                 {"stream_tail_fraction": 0.0}, {"stream_tail_fraction": 1.0}, {"stream_waves_per_cu": 4},
                 {"stream_waves_per_cu": 32}, {"timing_stride": 1}, {"timing_stride": 4}, {"stats": 1},
                 {"stream_tiles_per_chunk": 3, "stream_tail_fraction": 0.5, "stream_waves_per_cu": 9, "stats": 1}):
        got = run(opts)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), opts
    ctx = gpu_ctx_factory()
    for k, v in (("stream_tiles_per_chunk", 0), ("stream_tiles_per_chunk", 2.5), ("stream_tail_fraction", 1.5), ("stream_waves_per_cu", 33),
                 ("timing_stride", 0)):
        from cudaparticlesfoam_amd import _lib as L
        with pytest.raises(L.CpfError):
            ctx.set_option(k, v)
